#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, one bench line, one rocprofv3 kernel-trace summary.
# Usage: scripts/gpu_check.sh [tag]   -> everything lands under gpurun_out/<tag>/
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout ${TEST_TIMEOUT:-600} python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
  tail -5 "$OUT/pytest_gpu.log"
fi
if [ "${SKIP_BENCH:-0}" != 1 ]; then
  timeout ${BENCH_TIMEOUT:-400} python bench.py ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $?"; tail -c 3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
fi
if [ "${SKIP_PROF:-0}" != 1 ]; then
  export TMPDIR=/tmp
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d /tmp/prof -o $TAG --output-format csv -- \
      python "$ROOT/bench.py" ${PROF_ARGS:---steps 2 --warmup 1 --no-cpu-baseline} > "$OUT/prof_bench.json" 2> "$OUT/prof.err")
  echo "rocprof exit $?"
  find /tmp/prof -name '*stats*' -exec cp {} "$OUT/" \;
  ls -la /tmp/prof/* | head; ls -la "$OUT"
fi
if [ "${SKIP_REBAL:-0}" != 1 ]; then
  timeout ${REBAL_TIMEOUT:-400} python scripts/bench_rebalance.py ${REBAL_ARGS:---check} > "$OUT/bench_rebalance.json" 2> "$OUT/bench_rebalance.err"
  echo "rebalance bench exit $?"; tail -c 2000 "$OUT/bench_rebalance.json"; tail -5 "$OUT/bench_rebalance.err"
fi
