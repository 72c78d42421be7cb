#!/bin/bash
# The batched rank (cook_cycle_run_rank_multi) on the GPU: its tests, then A/B of the eight-pool cycle with and without it (COOK_RANK_BATCH),
# with the read-backs as recorded copies (COOK_BATCH_COPY_KERNEL=0), and at K = 1000.  Usage: scripts/r05_rank_batch_session.sh <tag>
set -u
TAG=${1:-r05rb}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=8
python scripts/kernel_rev.py | tee "$OUT/kernel_rev.txt"
( time timeout 420 python -m pytest tests/test_parity_gpu.py -x -q -k "rank_batch or timed_configuration or rank_parity or rank_equal or rank_tie or rank_golden or multi_pool or user_usage or considerable" ) > "$OUT/pytest_rank_batch.log" 2>&1
tail -5 "$OUT/pytest_rank_batch.log"
run() {  # name, env assignments..., then bench arguments after --
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 240 python bench.py --no-cpu-baseline --no-adjacent --no-extras --no-roofline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", "ms/cycle %.3f" % d["ms_per_step"], "p50 %.3f" % d["p50_cycle_latency_ms"], "phase", {k: round(v, 3) for k, v in d["phase_ms"].items()}, "parity", d.get("parity_checked"), "batch", d.get("rank_batch"))
except Exception as ex:
    print("$name", "FAILED", ex)
PY
}
for rep in 1 2; do
  run all_batch_$rep COOK_RANK_BATCH=1 -- --steps 20 --warmup 3
  run all_threads_$rep COOK_RANK_BATCH=0 -- --steps 20 --warmup 3
done
run all_batch_copies COOK_RANK_BATCH=1 COOK_BATCH_COPY_KERNEL=0 -- --steps 20 --warmup 3
run k1000_batch COOK_RANK_BATCH=1 -- --steps 60 --warmup 5 --considerable 1000
run k1000_threads COOK_RANK_BATCH=0 -- --steps 60 --warmup 5 --considerable 1000
COOK_SYNC_TRACE=0 python - <<'PY' > "$OUT/rank_only.txt" 2>&1
# the rank part alone: eight C4 pools, cook_cycle_run_rank_multi against a thread per pool
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from cook_amd import _abi as A, workload
from cook_amd.engine import Engine, cycle_run_rank_multi, cycle_match_multi
spec = workload.ClusterSpec()
pools = workload.make_pools(spec, range(spec.pools))
params = A.default_params(good_enough_fitness=1.0)
engines = [Engine(params, device=0) for _ in pools]
for e, p in zip(engines, pools.values()):
    e.cycle_stage(p.tasks, p.users, p.pending_jobs, p.offers, p.groups)
for K in (1000, 10 ** 9):
    for form in ("batch", "threads4", "serial"):
        tp = ThreadPoolExecutor(4)
        ts = []
        for it in range(12):
            t0 = time.perf_counter()
            if form == "batch":
                cycle_run_rank_multi(engines, K)
            elif form == "threads4":
                list(tp.map(lambda e: e.cycle_run_rank(K), engines))
            else:
                for e in engines:
                    e.cycle_run_rank(K)
            ts.append((time.perf_counter() - t0) * 1e3)
        st = engines[0].match_stats()
        print("K", K, form, "rank part of 8 pools: median %.3f ms, min %.3f" % (float(np.median(ts[2:])), min(ts)), {k: v for k, v in st.items() if k.startswith("rank_batch")} if form == "batch" else "")
for e in engines:
    e.close()
PY
cat "$OUT/rank_only.txt"
