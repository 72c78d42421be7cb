#!/bin/bash
# A tuning variant of the library next to the shipped one: scripts/build_variant.sh name -DCOOK_X=1 [-D...]  ->  cook_amd/libcookmatch_<name>.so
# (git-ignored like the shipped .so; scripts/served_probe.py and scripts/tune_run.sh take such names)
NAME=$1; shift
cd "$(dirname "$0")/.." || exit 1
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-fast-math -ffp-contract=off -fvisibility=hidden \
  -fvisibility-inlines-hidden -Wl,--version-script=cook_amd/csrc/exports.map -Wno-unused-function "$@" -o cook_amd/libcookmatch_$NAME.so cook_amd/csrc/engine.hip
