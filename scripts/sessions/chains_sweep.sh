#!/bin/bash
# how many launch chains for the 8 pools of one rank (COOK_MAX_CHAINS x pools in lockstep); COOK_MAX_RANK_CHAINS follows
mkdir -p gpurun_out/ch
for C in ${CHAINS:-2 4 8}; do
  COOK_MAX_CHAINS=$C timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-adjacent --no-check --no-roofline --no-extras > gpurun_out/ch/chains$C.json 2> gpurun_out/ch/chains$C.err
  python -c "import json;d=json.load(open('gpurun_out/ch/chains$C.json'));print('chains $C ms/cycle %.1f'%d['ms_per_step'], d['phase_ms'])"
done
