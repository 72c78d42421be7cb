#!/bin/bash
# GPU-box call: the K = 1000 cycle (config.clj:113) per match_algo, one pool and all eight.
set -u
TAG=${1:-k1000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
for S in ${STEPS:-one all}; do
  for ALGO in ${ALGOS:-0 4 6}; do
    if [ $S = one ]; then ARGS="--pools 1 --pending 125000 --running 50000 --offers 6250"; else ARGS=""; fi
    F="$OUT/${S}_a$ALGO"
    timeout 300 python bench.py $ARGS --considerable ${KK:-1000} --steps ${NSTEPS:-20} --warmup 3 --no-cpu-baseline --no-adjacent --no-extras --match-algo $ALGO ${BENCH_EXTRA:-} > "$F.json" 2> "$F.err"
    rc=$?
    python - <<PY
import json
try:
    t = open("$F.json").read()
    d = json.loads(t[t.index('{"metric'):])
    s = d["last_cycle"]["placement_stats_pool0"]
    print("$S algo $ALGO rc=$rc ms/cycle %.3f" % d["ms_per_step"], "parity", d.get("parity_checked"), "phase", {k: round(v, 3) for k, v in d["phase_ms"].items()}, "stage0", {k: round(v, 3) for k, v in d["last_cycle"]["stage_ms_pool0"].items()})
    print("    stats", {k: v for k, v in s.items() if v})
    ks = (d.get("roofline") or {}).get("kernels_ms_per_cycle", {})
    print("    kernels", {k: round(v, 3) for k, v in list(ks.items())[:12]})
except Exception as ex:
    print("$S algo $ALGO rc=$rc FAILED", ex)
PY
  done
done
