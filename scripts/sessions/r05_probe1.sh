#!/bin/bash
# one C4 pool (lockstep launches) under the library variants given: scripts/r05_probe1.sh tag lib1,lib2,... [pools8]
TAG=$1; LIBS=$2
OUT=gpurun_out/$TAG; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
PROBE_CYCLES=5 timeout 400 python scripts/served_probe.py $LIBS 'COOK_MATCH_SERVED=0' 1 2>&1 | grep -v amdgpu.ids > $OUT/probe1.txt
if [ -n "$3" ]; then PROBE_CYCLES=5 timeout 400 python scripts/served_probe.py $LIBS 'COOK_MATCH_SERVED=1' 8 2>&1 | grep -v amdgpu.ids > $OUT/probe8.txt; fi
cat $OUT/probe1.txt $OUT/probe8.txt 2>/dev/null
