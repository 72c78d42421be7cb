#!/bin/bash
# Round 4's evidence run on the GPU box (via gpurun): the GPU suite, the rocprofv3 passes of scripts/profile_round2.sh, the per-rank
# loads behind DESIGN.md's scaling table, a seeded fuzz sweep.  Everything lands under gpurun_out/<tag>/.
set -u
TAG=${1:-r04p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python scripts/kernel_rev.py > "$OUT/kernel_rev.txt"
timeout 600 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2
bash scripts/profile_round2.sh $TAG 2>&1 | tail -30
bash scripts/pools_per_gpu.sh > "$OUT/pools_per_gpu.txt" 2>&1; cat "$OUT/pools_per_gpu.txt"
timeout 600 python scripts/fuzz_sweep.py --match 150 --rebalance 40 --seed 404 > "$OUT/fuzz_gpu.txt" 2>&1; tail -2 "$OUT/fuzz_gpu.txt"
