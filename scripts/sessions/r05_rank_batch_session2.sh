#!/bin/bash
# Second look at the batched rank on the GPU: what a batch issues (COOK_BATCH_TRACE), per-kernel durations of the batched launches
# (rocprofv3 --kernel-trace --stats), the rank part alone in its three forms, the rest of the GPU tests of the batch.
set -u
TAG=${1:-r05rc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=8
python scripts/kernel_rev.py | tee "$OUT/kernel_rev.txt"
B8="--steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --no-adjacent --no-roofline"
COOK_BATCH_TRACE=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-extras --no-adjacent --no-roofline > /dev/null 2> "$OUT/batch_trace.err"
grep "^batch:" "$OUT/batch_trace.err" | sed 's/(grid.*//' | sort | uniq -c | sort -rn | head -60 > "$OUT/batch_trace_hist.txt"
head -30 "$OUT/batch_trace_hist.txt"
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/kt_b && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_b -o kt --output-format csv -- python "$ROOT/bench.py" $B8 > "$OUT/bench_under_rocprof_batch.json" 2> "$OUT/kt_batch.err"; find /tmp/kt_b -name '*kernel_stats*' -exec cp {} "$OUT/kernel_stats_batch.csv" \; )
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats_batch.csv")))
for r in rows[:40]:
    n = r["Name"]
    short = n.split("cook_multi<&")[1].split("(")[0] if "cook_multi<&" in n else n.split("(")[0]
    print("%-50s calls %5s avg %9.1f us total %9.1f us" % (short[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
python - <<'PY' > "$OUT/rank_only.txt" 2>&1
# the rank part alone: eight C4 pools, cook_cycle_run_rank_multi against a thread per pool
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from cook_amd import _abi as A, workload
from cook_amd.engine import Engine, cycle_run_rank_multi
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
  run all_batch1_$rep COOK_RANK_BATCHES=1 -- --steps 20 --warmup 3
  run all_batch2_$rep COOK_RANK_BATCHES=2 -- --steps 20 --warmup 3
  run all_threads_$rep COOK_RANK_BATCH=0 -- --steps 20 --warmup 3
done
run all_batch4 COOK_RANK_BATCHES=4 -- --steps 20 --warmup 3
run all_batch1_copies COOK_BATCH_COPY_KERNEL=0 -- --steps 20 --warmup 3
run k1000_batch1 COOK_RANK_BATCHES=1 -- --steps 60 --warmup 5 --considerable 1000
run k1000_batch2 COOK_RANK_BATCHES=2 -- --steps 60 --warmup 5 --considerable 1000
run k1000_threads COOK_RANK_BATCH=0 -- --steps 60 --warmup 5 --considerable 1000
( time timeout 420 python -m pytest tests/test_parity_gpu.py -q -k "rank_batch or timed_configuration or multi_pool or considerable or user_usage" ) > "$OUT/pytest_rank_batch.log" 2>&1
tail -5 "$OUT/pytest_rank_batch.log"
