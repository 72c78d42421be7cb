#!/bin/bash
# GPU-box call for match_v3 tuning: one C4 pool (and optionally all eight) with match_algo 6 for each library in LIBS x each look-ahead in LAS.
set -u
TAG=${1:-v3s}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
ONE="--pools 1 --pending 125000 --running 50000 --offers 6250 --steps 3 --warmup 1 --no-cpu-baseline --no-adjacent --no-extras"
ALL="--steps 3 --warmup 1 --no-cpu-baseline --no-adjacent --no-extras"
if [ -n "${PYTEST_K:-}" ]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "$PYTEST_K" > "$OUT/pytest.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
fi
for S in ${STEPS:-one}; do
  for LIB in ${LIBS:-default}; do
    NAME=$(basename $LIB .so)
    if [ "$LIB" = default ]; then unset COOK_LIB; else export COOK_LIB=$ROOT/$LIB; fi
    for LA in ${LAS:-32}; do
      export COOK_V3_LA=$LA
      if [ $S = one ]; then ARGS="$ONE"; else ARGS="$ALL"; fi
      F="$OUT/${S}_${NAME}_la$LA"
      timeout 300 python bench.py $ARGS --match-algo 6 ${BENCH_EXTRA:-} > "$F.json" 2> "$F.err"
      rc=$?
      python - <<PY
import json
try:
    t = open("$F.json").read()
    d = json.loads(t[t.index('{"metric'):])
    s = d["last_cycle"]["placement_stats_pool0"]
    keep = ("rounds", "stop_list", "stop_full", "v3_generations", "v3_walked", "v3_settled", "v3_scan_steps", "v3_visits", "v3_opens", "v3_fast", "v3_total_us", "v3_regen_us", "v3_wait_us", "v3_refused")
    print("$NAME $S la=$LA rc=$rc ms/cycle %.2f" % d["ms_per_step"], "matched", d["last_cycle"]["matched"], "parity", d.get("parity_checked"), {k: s.get(k) for k in keep})
except Exception as ex:
    print("$NAME $S la=$LA rc=$rc FAILED", ex)
PY
    done
  done
done
