#!/bin/bash
# Shader clock under the placement's load (one wave per pool busy: does the power governor keep the clock down?) and the cycle
# time with the performance level forced to high.  usage: scripts/gpu_clocks.sh tag
TAG=${1:-clocks}; OUT=gpurun_out/$TAG; mkdir -p $OUT
rocm-smi --showclocks --showperflevel > $OUT/idle.txt 2>&1
B="--steps 20 --warmup 2 --no-cpu-baseline --no-adjacent --no-extras --no-check"
(python bench.py $B > $OUT/auto.json 2> $OUT/auto.err) &
BP=$!
sleep 14; rocm-smi --showclocks > $OUT/during_auto.txt 2>&1
wait $BP
rocm-smi --setperflevel high > $OUT/set_high.txt 2>&1
rocm-smi --showclocks --showperflevel > $OUT/after_set.txt 2>&1
(python bench.py $B > $OUT/high.json 2> $OUT/high.err) &
BP=$!
sleep 14; rocm-smi --showclocks > $OUT/during_high.txt 2>&1
wait $BP
rocm-smi --setperflevel auto >> $OUT/set_high.txt 2>&1
python - <<PY
import json
for k in ("auto","high"):
    try:
        d=json.loads(open("$OUT/%s.json"%k).read().strip().splitlines()[-1]); print(k,"ms/cycle %.2f"%d["ms_per_step"], d["phase_ms"])
    except Exception as ex: print(k,"FAILED",ex)
PY
grep -h -i "sclk\|perf" $OUT/idle.txt $OUT/during_auto.txt $OUT/after_set.txt $OUT/during_high.txt | head -20
