#!/bin/bash
# the cycle_update / boundary leg: GPU tests of the update path, then the bench's boundary object (no CPU baseline, no extras)
T=${1:-bd}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_replay.py -x -q -m gpu -k "update or replay or cycle" > gpurun_out/$T/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/$T/pytest.log | tail -2
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --boundary --no-adjacent --no-check > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/$T/bench.json')); print(round(d['ms_per_step'],2), json.dumps(d['boundary'])[:900])
PY
