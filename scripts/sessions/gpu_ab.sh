#!/bin/bash
# A/B of library builds on the GPU box: one C4 pool alone and the 8-pool cluster, parity-checked, for each library given.
# Usage: scripts/gpu_ab.sh tag lib1 [lib2 ...]   ("default" = cook_amd/libcookmatch.so)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for LIB in "$@"; do
  NAME=$(basename $LIB .so)
  if [ "$LIB" = default ]; then unset COOK_LIB; else export COOK_LIB=$PWD/$LIB; fi
  timeout 200 python bench.py --pools 1 --pending 125000 --running 50000 --offers 6250 --steps 8 --warmup 2 --no-cpu-baseline --no-adjacent --no-extras > $OUT/one_$NAME.json 2> $OUT/one_$NAME.err
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-adjacent --no-extras > $OUT/all_$NAME.json 2> $OUT/all_$NAME.err
  python - <<PY
import json
for k in ("one", "all"):
    try:
        d = json.loads(open("$OUT/%s_$NAME.json" % k).read().strip().splitlines()[-1])
        s = d["last_cycle"]["placement_stats_pool0"]
        print("$NAME", k, "ms/cycle %.2f" % d["ms_per_step"], "parity", d.get("parity_checked"), "rounds", s["rounds"], "matched", d["last_cycle"]["matched"],
              "setup_us", s["setup_us"], "seq_us", s["seq_us"], {k2: round(v, 1) for k2, v in list(d["roofline"]["kernels_ms_per_cycle"].items())[:3]})
    except Exception as ex:
        print("$NAME", k, "FAILED", ex)
PY
done
