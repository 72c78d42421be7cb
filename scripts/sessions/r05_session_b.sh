#!/bin/bash
# round 5, second GPU session: window / list-length variants under served walkers, the guarded fuzz sweep's rate, the update's outliers
OUT=gpurun_out/r05y; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
LIBS=default,libcookmatch_we1472.so,libcookmatch_we1920.so,libcookmatch_lm64.so,libcookmatch_we1920lm64.so
PROBE_CYCLES=5 timeout 400 python scripts/served_probe.py $LIBS 'COOK_MATCH_SERVED=1' 8 > $OUT/probe8.txt 2>&1
PROBE_CYCLES=5 timeout 300 python scripts/served_probe.py $LIBS 'COOK_MATCH_SERVED=0' 1 > $OUT/probe1.txt 2>&1
( time timeout 300 python scripts/fuzz_sweep.py --guard --match 100 --rebalance 40 --multi 20 --seed 777 ) > $OUT/fuzz_guard.txt 2>&1
timeout 200 python scripts/update_outliers.py > $OUT/update_outliers.txt 2>&1
tail -n 30 $OUT/probe8.txt $OUT/probe1.txt $OUT/fuzz_guard.txt $OUT/update_outliers.txt
