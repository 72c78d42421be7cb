#!/bin/bash
OUT=gpurun_out/r05z; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
LIBS=default,libcookmatch_lm64.so,libcookmatch_lm56.so,libcookmatch_lm64l12.so
PROBE_CYCLES=5 timeout 400 python scripts/served_probe.py $LIBS 'COOK_MATCH_SERVED=1;COOK_MATCH_SERVED=1 COOK_EVAL_SPLIT=2;COOK_MATCH_SERVED=1 COOK_EVAL_SPLIT=4' 8 > $OUT/probe8.txt 2>&1
PROBE_CYCLES=5 timeout 300 python scripts/served_probe.py $LIBS 'COOK_MATCH_SERVED=0' 1 > $OUT/probe1.txt 2>&1
for L in default lm64 lm64l12; do
  if [ $L = default ]; then unset COOK_LIB; else export COOK_LIB=$PWD/cook_amd/libcookmatch_$L.so; fi
  echo "== $L"; timeout 200 python scripts/c3_probe.py
done > $OUT/c23.txt 2>&1
unset COOK_LIB
( time timeout 600 python scripts/fuzz_sweep.py --guard --match 1400 --rebalance 400 --multi 250 --seed 20260924 ) > $OUT/fuzz_guard_2050.txt 2>&1
tail -n 30 $OUT/probe8.txt $OUT/probe1.txt $OUT/c23.txt $OUT/fuzz_guard_2050.txt
