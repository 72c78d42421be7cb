#!/bin/bash
# bench.py's boundary leg N times under an environment setting: how many of the 50 update samples per run are slow, and where (boundary.update_outliers)
# usage: scripts/update_outlier_hunt.sh tag runs [ENV=VALUE ...]
TAG=$1; RUNS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
for i in $(seq 1 $RUNS); do
  timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-adjacent --boundary > $OUT/b_$i.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$OUT/b_$i.json"))["boundary"]
print("$* run $i: median %.3f max %.3f outliers %s" % (d["update_ms"], d["update_ms_max"], [(o["sample"], o["ms"], o["per_pool_slowest_phase"], o["per_pool_slowest_phase_us"]) for o in d["update_outliers"]]))
PY
done
