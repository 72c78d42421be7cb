#!/bin/bash
OUT=gpurun_out/r05za; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
for GE in 1.0 0.8; do
for L in default lm64 lm64ge48 lm64ge64; do
  if [ $L = default ]; then unset COOK_LIB; else export COOK_LIB=$PWD/cook_amd/libcookmatch_$L.so; fi
  echo "== $L ge $GE"; PROBE_GE=$GE timeout 200 python scripts/c3_probe.py 2>&1 | grep -v amdgpu.ids
done; done > $OUT/c23.txt 2>&1
unset COOK_LIB
LIBS=default,libcookmatch_lm64ge48.so,libcookmatch_lm64ge64.so
PROBE_GE=0.8 PROBE_CYCLES=4 timeout 400 python scripts/served_probe.py $LIBS 'COOK_MATCH_SERVED=1' 8 > $OUT/probe8_ge08.txt 2>&1
PROBE_GE=0.8 PROBE_CYCLES=4 timeout 300 python scripts/served_probe.py $LIBS 'COOK_MATCH_SERVED=0' 1 > $OUT/probe1_ge08.txt 2>&1
tail -n 30 $OUT/c23.txt $OUT/probe8_ge08.txt $OUT/probe1_ge08.txt
