#!/bin/bash
# One GPU-box call of round 3 (via gpurun): the GPU suite, the driver's bench line and the rocprofv3 evidence of the SAME binary,
# everything under gpurun_out/<tag>/ together with the git revision of the kernel sources it was taken at.
#   STEPS: tests bench prof (default all three); PROF_STEPS is handed to scripts/profile_round2.sh (kt8 kt1 pmc8 sq1 ktrb)
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python scripts/kernel_rev.py > "$OUT/kernel_rev.txt" 2>&1
for S in ${STEPS:-tests bench prof}; do
  case $S in
    tests)
      timeout ${TEST_TIMEOUT:-1200} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.log" 2>&1
      echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -5 "$OUT/pytest_gpu.log";;
    bench)
      timeout ${BENCH_TIMEOUT:-900} python bench.py ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
      echo "bench exit $?"; tail -c 1500 "$OUT/bench.json"; tail -3 "$OUT/bench.err";;
    prof)
      STEPS="${PROF_STEPS:-kt8 kt1 pmc8 sq1}" bash scripts/profile_round2.sh "$TAG";;
  esac
done
ls "$OUT"
