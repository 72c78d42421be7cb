#!/bin/bash
# quick look on the GPU: the rank / cycle parity tests, one pool alone and the 8-pool cycle (no CPU baseline), rank kernel stats
T=${1:-quick}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "rank or cycle" > gpurun_out/$T/pytest_rank.log 2>&1; tail -3 gpurun_out/$T/pytest_rank.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-adjacent --as-rank-of 8 > gpurun_out/$T/one_pool.json 2> gpurun_out/$T/one_pool.err
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-adjacent > gpurun_out/$T/eight.json 2> gpurun_out/$T/eight.err
python - <<PY
import json
for n in ('one_pool','eight'):
    try:
        d=json.load(open('gpurun_out/$T/%s.json'%n)); print(n, round(d['ms_per_step'],2), d['phase_ms'], d.get('parity_checked'))
    except Exception as ex: print(n, 'failed', ex)
PY
