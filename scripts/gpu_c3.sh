#!/bin/bash
# GPU-box call: BASELINE.json configs[2] (one pool, 200k pending x 20k offers, constraints) for each library in LIBS.
set -u
TAG=${1:-c3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
for LIB in ${LIBS:-default}; do
  NAME=$(basename $LIB .so)
  if [ "$LIB" = default ]; then unset COOK_LIB; else export COOK_LIB=$ROOT/$LIB; fi
  F="$OUT/c3_$NAME"
  timeout 600 python bench.py --pools 1 --pending 200000 --running 80000 --offers 20000 --users 2000 --steps 3 --warmup 1 --no-cpu-baseline --no-adjacent --no-extras ${BENCH_EXTRA:-} > "$F.json" 2> "$F.err"
  rc=$?
  python - <<PY
import json
try:
    t = open("$F.json").read()
    d = json.loads(t[t.index('{"metric'):])
    s = d["last_cycle"]["placement_stats_pool0"]
    print("$NAME rc=$rc ms/cycle %.2f" % d["ms_per_step"], "matched", d["last_cycle"]["matched"], "parity", d.get("parity_checked"), "stage0", {k: round(v, 2) for k, v in d["last_cycle"]["stage_ms_pool0"].items()})
    print("    stats", {k: v for k, v in s.items() if v})
    ks = (d.get("roofline") or {}).get("kernels_ms_per_cycle", {})
    print("    kernels", {k: round(v, 2) for k, v in list(ks.items())[:6]})
except Exception as ex:
    print("$NAME rc=$rc FAILED", ex)
PY
done
