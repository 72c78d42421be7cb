#!/bin/bash
# 8-pool cycle time under environment variants (launch chains, hardware queues, rank / placement barrier)
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-adjacent --no-extras --no-check"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['phase_ms'])"; }
for v in "$@"; do run $v; done
