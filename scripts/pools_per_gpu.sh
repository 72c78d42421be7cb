#!/bin/bash
# cycle time of the pools ONE rank of an N-GPU job holds (8 / 4 / 2 / 1 pools of the fixed 8-pool cluster), measured on one GPU
for N in 1 2 4 8; do
  python bench.py --as-rank-of $N --steps 6 --warmup 2 \
      --no-cpu-baseline --no-roofline --no-adjacent --no-extras --no-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); s = d['last_cycle']
print('N', $N, 'ms', round(d['ms_per_step'], 2), 'phases', {k: round(v, 2) for k, v in d['phase_ms'].items()}, 'pool0', s['stage_ms_pool0'], 'rounds', s['placement_stats_pool0']['rounds'])"
done
