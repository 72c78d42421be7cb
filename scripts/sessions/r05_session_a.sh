#!/bin/bash
# round 5, first GPU session: the served walkers LIVE (walker launch beside serve launches) — parity of the multi-pool tests and of the
# timed configuration, then A/B timings against the lockstep chains on the same box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05a
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=8
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "multi_pool or lockstep or many_pools or timed_configuration" > "$OUT/pytest_served.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_served.log"; tail -8 "$OUT/pytest_served.log"
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-adjacent --no-extras"
run() {  # name, env..., -- bench args
  local name=$1; shift
  ( for kv in "$@"; do [ "$kv" = "--" ] && break; export "$kv"; done
    while [ "$1" != "--" ]; do shift; done; shift
    timeout 300 python bench.py $ARGS "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
    echo "$name exit $?" )
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$name.json"))
    s = d["last_cycle"]["placement_stats_pool0"]
    print("$name ms/cycle %.2f" % d["ms_per_step"], "parity", d["parity_checked"], "phase", {k: round(v, 2) for k, v in d["phase_ms"].items()},
          {k: v for k, v in s.items() if k in ("rounds", "setup_us", "seq_us", "served_mode", "serve_iterations", "serve_empty_iterations", "serve_pool_windows", "serve_latch_wait_us", "served_fell_back")},
          {k2: v for k2, v in list((d.get("roofline") or {}).get("kernels_ms_per_cycle", {}).items())[:5]})
except Exception as ex:
    print("$name FAILED", ex)
PY
  tail -3 "$OUT/$name.err"
}
run served8 COOK_MATCH_SERVED=1 --
run lockstep8 COOK_MATCH_SERVED=0 --
run served1 COOK_MATCH_SERVED=1 -- --as-rank-of 8 --no-check
run lockstep1 COOK_MATCH_SERVED=0 -- --as-rank-of 8 --no-check
run served8_poll10 COOK_MATCH_SERVED=1 COOK_SERVE_POLL_US=10 -- --no-check
run served8_step COOK_MATCH_SERVED=1 COOK_SERVE_STEP=1 -- --no-check --steps 2
