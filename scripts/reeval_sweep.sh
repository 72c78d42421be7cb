#!/bin/bash
# tuning run: bounded in-place re-evaluations per round (COOK_REEVAL_MAX) on one C4 pool and on the 8-pool cluster
cd ${GRAFT_REPO_ROOT:-.}
for n in ${SWEEP:-0 1 2 4 8}; do
  export COOK_REEVAL_MAX=$n
  [ "$n" = 0 ] && unset COOK_REEVAL_MAX
  STEPS="${SWEEP_STEPS:-one all}" bash scripts/gpu_session.sh rv$n 2>&1 | grep "ms/cycle" | sed "s/^/reeval_max=$n /"
done
