#!/bin/bash
# Per-category cycle counts of the placement walk (-DCOOK_WALK_PROF builds, one C4 pool): scripts/gpu_walkprof.sh tag lib1 [lib2 ...]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for LIB in "$@"; do
  NAME=$(basename $LIB .so)
  COOK_LIB=$PWD/$LIB timeout 200 python bench.py --pools 1 --pending 125000 --running 50000 --offers 6250 --steps 2 --warmup 1 --no-cpu-baseline --no-adjacent --no-extras --no-check > $OUT/$NAME.json 2> $OUT/$NAME.err
  echo "$NAME: $(grep WALKPROF $OUT/$NAME.err | tail -1)"
done
