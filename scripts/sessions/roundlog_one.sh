#!/bin/bash
# per-round log of one C4 pool (launch path), with and without long windows
mkdir -p gpurun_out/rl3
for W in 1 0; do
  COOK_WLONG=$W COOK_ROUND_LOG=$PWD/gpurun_out/rl3/rounds_wlong$W.csv timeout 200 python bench.py --pools 1 --pending 125000 --running 50000 --offers 6250 --steps 2 --warmup 1 --no-adjacent --no-cpu-baseline --no-check > gpurun_out/rl3/bench$W.json 2> gpurun_out/rl3/bench$W.err
  echo "wlong $W rc=$?"; python scripts/round_log_summary.py gpurun_out/rl3/rounds_wlong$W.csv | head -8
done
