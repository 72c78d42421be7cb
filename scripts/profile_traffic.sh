#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace summary + the two PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
# gfx950: 3 + 2 TCC slots of 4, MI355X_MICROARCH.md "rocprofv3 PMC slots") of the SAME bench command.  Counter passes carry no
# trace domain other than the kernel list (gpurun refuses pmc + sys/hip traces).  Output: gpurun_out/<tag>/.
set -u
TAG=${1:-r01prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=${PROF_ARGS:---steps 2 --warmup 1 --no-cpu-baseline}
cd /tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python "$ROOT/bench.py" $ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
find /tmp/kt -name '*kernel_stats*' -exec cp {} "$OUT/kernel_stats.csv" \;
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2> "$OUT/pmc_$c.err"
  python "$ROOT/scripts/pmc_summary.py" /tmp/pmc_$c 40 > "$OUT/pmc_$c.txt"
  cat "$OUT/pmc_$c.txt" | head -5
done
head -6 "$OUT/kernel_stats.csv" | cut -c1-160
