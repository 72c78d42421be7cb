#!/bin/bash
# per-dispatch trace of the bench command at the shipped revision: the rank phase's batched launches (scripts/trace_rank_batches.py) and
# the served cycle's overlap (scripts/trace_overlap.py).  Usage: scripts/r05_rank_trace_session.sh <tag>
set -u
TAG=${1:-r05zh}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
python scripts/kernel_rev.py | tee "$OUT/kernel_rev.txt"
B8="--steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extras --no-adjacent --no-roofline"
for v in 4 1; do
  ( cd /tmp && rm -rf /tmp/kt_r$v && COOK_RANK_BATCHES=$v timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_r$v -o kt --output-format csv -- python "$ROOT/bench.py" $B8 > "$OUT/bench_under_trace_$v.json" 2> "$OUT/kt_$v.err" )
  python scripts/trace_rank_batches.py /tmp/kt_r$v "$OUT/rank_batches_trace_$v.txt" | head -30
done
python scripts/trace_overlap.py /tmp/kt_r4 "$OUT/kernel_trace_overlap.txt" | head -8
