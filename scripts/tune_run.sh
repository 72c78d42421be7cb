#!/bin/bash
# Tuning runs on the GPU box: the bench (one pool alone, then the 8-pool cluster) for each library variant given.
# Usage: scripts/tune_run.sh tag lib1 [lib2 ...]   ("default" = cook_amd/libcookmatch.so)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for LIB in "$@"; do
  NAME=$(basename $LIB .so)
  if [ "$LIB" = default ]; then unset COOK_LIB; else export COOK_LIB=$PWD/$LIB; fi
  timeout 120 python bench.py --pools 1 --pending 125000 --running 50000 --offers 6250 --steps 4 --warmup 1 --no-cpu-baseline --no-adjacent > $OUT/one_$NAME.json 2> $OUT/one_$NAME.err
  timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-adjacent > $OUT/all_$NAME.json 2> $OUT/all_$NAME.err
  python - <<PY
import json
for k in ("one", "all"):
    try:
        d = json.load(open("$OUT/%s_$NAME.json" % k))
        s = d["last_cycle"]["placement_stats_pool0"]
        print("$NAME", k, "ms/cycle %.2f" % d["ms_per_step"], "rounds", s["rounds"], "matched", d["last_cycle"]["matched"], "setup_us", s["setup_us"], "seq_us", s["seq_us"],
              {k2: v for k2, v in list(d["roofline"]["kernels_ms_per_cycle"].items())[:3]})
    except Exception as ex:
        print("$NAME", k, "FAILED", ex)
PY
done
