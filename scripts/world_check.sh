#!/bin/bash
# one C4 pool through match_world for each library given, verified against the oracle (bench.py's parity check)
for LIB in "$@"; do
  if [ "$LIB" = default ]; then unset COOK_LIB; else export COOK_LIB=$PWD/$LIB; fi
  timeout 200 python bench.py --match-algo 5 --pools 1 --pending 125000 --running 50000 --offers 6250 --steps 3 --warmup 1 --no-adjacent --no-roofline > /tmp/wc.json 2> /tmp/wc.err
  echo "$LIB rc=$? $(python -c "import json;d=json.load(open('/tmp/wc.json'));print('ms',round(d['ms_per_step'],1),'parity',d['parity_checked'],d['last_cycle']['placement_stats_pool0']['rounds'],d['last_cycle']['matched'])" 2>/dev/null) $(grep -m1 PARITY /tmp/wc.err)"
done
