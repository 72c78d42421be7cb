#!/bin/bash
# A/B of environment settings on the shipped library: scripts/r05_ab_env.sh tag 'SETTINGS;SETTINGS;...' (one pool in lockstep launches, then eight served)
TAG=$1; SET=$2
OUT=gpurun_out/$TAG; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
S1=$(echo "$SET" | sed 's/\([^;]*\)/COOK_MATCH_SERVED=0 \1/g')
S8=$(echo "$SET" | sed 's/\([^;]*\)/COOK_MATCH_SERVED=1 \1/g')
PROBE_CYCLES=5 timeout 400 python scripts/served_probe.py ${LIBS:-default} "$S1" 1 2>&1 | grep -v amdgpu.ids > $OUT/probe1.txt
PROBE_CYCLES=5 timeout 400 python scripts/served_probe.py ${LIBS:-default} "$S8" 8 2>&1 | grep -v amdgpu.ids > $OUT/probe8.txt
cat $OUT/probe1.txt $OUT/probe8.txt
