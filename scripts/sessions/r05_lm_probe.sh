#!/bin/bash
# merged candidate lists of 64 entries instead of 48 (libcookmatch_lm64.so: scripts/build_variant.sh lm64 -DCOOK_MV_LM=64) at the operating points the
# first session did not measure it at: K = 1000 (the reference's default cap), next to the eight-pool cycle and one pool alone
set -u
TAG=${1:-r05zk}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=8
for LIB in default lm64 default lm64; do
  if [ "$LIB" = default ]; then unset COOK_LIB; else export COOK_LIB=$PWD/cook_amd/libcookmatch_$LIB.so; fi
  for CFG in "k1000 --steps 80 --warmup 5 --considerable 1000" "all --steps 16 --warmup 3" "one --pools 1 --pending 125000 --running 50000 --offers 6250 --steps 10 --warmup 2"; do
    set -- $CFG; NAME=$1; shift
    timeout 200 python bench.py --no-cpu-baseline --no-adjacent --no-extras --no-roofline "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['last_cycle']['placement_stats_pool0']
print('$LIB', '$NAME', 'ms/cycle %.3f' % d['ms_per_step'], 'phase', {k: round(v, 3) for k, v in d['phase_ms'].items()}, 'parity', d.get('parity_checked'), 'rounds', s['rounds'], 'segments', s['segments'], 'visited', s['visited'])" | tee -a "$OUT/lm_probe.txt"
  done
done
