#!/bin/bash
# GPU-box call for match_v3 work: the v3 tests, then one pool and eight pools with match_algo 6 for each library in LIBS.
set -u
TAG=${1:-v3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python scripts/kernel_rev.py > "$OUT/kernel_rev.txt" 2>&1
ONE="--pools 1 --pending 125000 --running 50000 --offers 6250 --steps 4 --warmup 1 --no-cpu-baseline --no-adjacent --no-extras"
ALL="--steps 4 --warmup 1 --no-cpu-baseline --no-adjacent --no-extras"
for S in ${STEPS:-tests one all}; do
  case $S in
    tests)
      timeout ${TEST_TIMEOUT:-900} python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "${PYTEST_K:-v3}" > "$OUT/pytest_v3.log" 2>&1
      echo "pytest exit $?" >> "$OUT/pytest_v3.log"; tail -4 "$OUT/pytest_v3.log";;
    one|all)
      for LIB in ${LIBS:-default}; do
        NAME=$(basename $LIB .so)
        if [ "$LIB" = default ]; then unset COOK_LIB; else export COOK_LIB=$ROOT/$LIB; fi
        for ALGO in ${ALGOS:-6}; do
          if [ $S = one ]; then ARGS="$ONE"; else ARGS="$ALL"; fi
          timeout 300 python bench.py $ARGS --match-algo $ALGO ${BENCH_EXTRA:-} > "$OUT/${S}_${NAME}_a$ALGO.json" 2> "$OUT/${S}_${NAME}_a$ALGO.err"
          echo "$S $NAME algo $ALGO exit $?"; tail -2 "$OUT/${S}_${NAME}_a$ALGO.err"
          python - <<PY
import json
try:
    t = open("$OUT/${S}_${NAME}_a$ALGO.json").read()
    d = json.loads(t[t.index('{"metric'):])
    s = d["last_cycle"]["placement_stats_pool0"]
    print("$NAME $S algo $ALGO ms/cycle %.2f" % d["ms_per_step"], "matched", d["last_cycle"]["matched"], "parity", d.get("parity_checked"), "phase", {k: round(v, 2) for k, v in d["phase_ms"].items()},
          "stage0", {k: round(v, 2) for k, v in d["last_cycle"]["stage_ms_pool0"].items()})
    print("   stats", {k: v for k, v in s.items() if v})
    print("   kernels", {k2: v for k2, v in list((d.get("roofline") or {}).get("kernels_ms_per_cycle", {}).items())[:5]})
except Exception as ex:
    print("$NAME $S algo $ALGO FAILED", ex)
PY
        done
      done
      unset COOK_LIB;;
  esac
done
