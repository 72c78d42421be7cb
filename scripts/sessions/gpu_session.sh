#!/bin/bash
# One GPU-box session (via gpurun): steps selected by STEPS (space separated), everything lands under gpurun_out/<tag>/.
#   ubench   scripts/ubench_wave (single-wave latency table)
#   tests    pytest -m gpu
#   one      bench.py, one C4 pool alone, for each library in LIBS (default = cook_amd/libcookmatch.so)
#   all      bench.py, the 8-pool cluster, for each library in LIBS
#   bench    the driver's bench line (python bench.py $BENCH_ARGS)
#   prof     rocprofv3 --kernel-trace --stats of bench.py
set -u
TAG=${1:-s}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
STEPS=${STEPS:-"ubench tests one all"}
LIBS=${LIBS:-default}
ONE_ARGS="${ONE_PRE:-} --pools 1 --pending 125000 --running 50000 --offers 6250 --steps 4 --warmup 1 --no-cpu-baseline --no-adjacent --no-check"
ALL_ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-adjacent --no-check"
for S in $STEPS; do
  case $S in
    ubench)
      [ -x scripts/ubench_wave ] || hipcc -O3 --offload-arch=gfx950 -o scripts/ubench_wave scripts/ubench_wave.hip
      timeout 60 scripts/ubench_wave > "$OUT/ubench.json" 2> "$OUT/ubench.err"; echo "ubench exit $?"; cat "$OUT/ubench.json";;
    tests)
      timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.log" 2>&1
      echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -6 "$OUT/pytest_gpu.log";;
    one|all)
      for LIB in $LIBS; do
        NAME=$(basename $LIB .so)
        if [ "$LIB" = default ]; then unset COOK_LIB; else export COOK_LIB=$ROOT/$LIB; fi
        if [ $S = one ]; then ARGS="$ONE_ARGS ${ONE_EXTRA:-}"; else ARGS="$ALL_ARGS ${ALL_EXTRA:-}"; fi
        timeout 180 python bench.py $ARGS > "$OUT/${S}_$NAME.json" 2> "$OUT/${S}_$NAME.err"
        echo "$S $NAME exit $?"; grep -E "WALKPROF|WORLDPROF" "$OUT/${S}_$NAME.err" | tail -4
        python - <<PY
import json
try:
    d = json.load(open("$OUT/${S}_$NAME.json"))
    s = d["last_cycle"]["placement_stats_pool0"]
    print("$NAME $S ms/cycle %.2f" % d["ms_per_step"], "rounds", s.get("rounds"), "matched", d["last_cycle"]["matched"], "setup_us", s.get("setup_us"), "seq_us", s.get("seq_us"),
          {k2: v for k2, v in list((d.get("roofline") or {}).get("kernels_ms_per_cycle", {}).items())[:4]})
except Exception as ex:
    print("$NAME $S FAILED", ex)
PY
      done
      unset COOK_LIB;;
    bench)
      timeout ${BENCH_TIMEOUT:-600} python bench.py ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
      echo "bench exit $?"; tail -c 2500 "$OUT/bench.json"; tail -5 "$OUT/bench.err";;
    prof)
      export TMPDIR=/tmp
      rm -rf /tmp/prof && mkdir -p /tmp/prof
      (cd /tmp && timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d /tmp/prof -o $TAG --output-format csv -- \
          python "$ROOT/bench.py" ${PROF_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-adjacent --no-check} > "$OUT/prof_bench.json" 2> "$OUT/prof.err")
      echo "rocprof exit $?"
      find /tmp/prof -name '*stats*' -exec cp {} "$OUT/" \;
      ls "$OUT";;
  esac
done
