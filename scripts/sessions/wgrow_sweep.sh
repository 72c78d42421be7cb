#!/bin/bash
# window growth factor (COOK_WGROW_PCT: next window = that percentage of what the last round resolved) for the 8-pool cycle and one pool alone
mkdir -p gpurun_out/wg
B="--steps 4 --warmup 1 --no-cpu-baseline --no-adjacent --no-check --no-roofline --no-extras"
for W in ${WGROWS:-105 120 150 200 300}; do
  COOK_WGROW_PCT=$W timeout 200 python bench.py $B > gpurun_out/wg/all$W.json 2> gpurun_out/wg/all$W.err
  COOK_WGROW_PCT=$W timeout 200 python bench.py $B --pools 1 --pending 125000 --running 50000 --offers 6250 > gpurun_out/wg/one$W.json 2> gpurun_out/wg/one$W.err
  python -c "
import json
a=json.load(open('gpurun_out/wg/all$W.json')); o=json.load(open('gpurun_out/wg/one$W.json'))
print('wgrow $W: 8 pools %.1f ms (rounds %d)  one pool %.1f ms (rounds %d, visited %d)' % (a['ms_per_step'], a['last_cycle']['placement_stats_pool0']['rounds'], o['ms_per_step'], o['last_cycle']['placement_stats_pool0']['rounds'], o['last_cycle']['placement_stats_pool0']['visited']))"
done
