#!/bin/bash
# GPU-box call: good-enough-fitness 0.8 (config.clj:111) with and without the resolve kernel's good-enough fast path.
set -u
TAG=${1:-ge}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ -n "${PYTEST_K:-}" ]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "$PYTEST_K" > "$OUT/pytest.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
fi
for S in ${STEPS:-one all}; do
  for FAST in ${FASTS:-1 0}; do
    for GE in ${GES:-0.8}; do
      if [ $S = one ]; then ARGS="--pools 1 --pending 125000 --running 50000 --offers 6250"; else ARGS=""; fi
      F="$OUT/${S}_ge${GE}_fast$FAST"
      COOK_GE_FAST=$FAST timeout 300 python bench.py $ARGS --good-enough $GE --steps 3 --warmup 1 --no-cpu-baseline --no-adjacent --no-extras > "$F.json" 2> "$F.err"
      rc=$?
      python - <<PY
import json
try:
    t = open("$F.json").read()
    d = json.loads(t[t.index('{"metric'):])
    s = d["last_cycle"]["placement_stats_pool0"]
    print("$S ge=$GE fast=$FAST rc=$rc ms/cycle %.2f" % d["ms_per_step"], "matched", d["last_cycle"]["matched"], "parity", d.get("parity_checked"), "stage0", {k: round(v, 2) for k, v in d["last_cycle"]["stage_ms_pool0"].items()})
    print("    stats", {k: v for k, v in s.items() if v})
except Exception as ex:
    print("$S ge=$GE fast=$FAST rc=$rc FAILED", ex)
PY
    done
  done
done
