#!/bin/bash
# per-round logs of one C4 pool: launches (algo 2) against match_world (algo 5), to find the first round that differs
mkdir -p gpurun_out/rl2
for A in 2 5; do
  COOK_ROUND_LOG=$PWD/gpurun_out/rl2/rounds_algo$A.csv timeout 200 python bench.py --match-algo $A --pools 1 --pending 125000 --running 50000 --offers 6250 --steps 1 --warmup 0 --no-adjacent --no-roofline --no-cpu-baseline --no-check > gpurun_out/rl2/bench$A.json 2> gpurun_out/rl2/bench$A.err
  echo "algo $A rc=$?"; wc -l gpurun_out/rl2/rounds_algo$A.csv
done
python - <<'PY'
a=open('gpurun_out/rl2/rounds_algo2.csv').read().split('\n'); b=open('gpurun_out/rl2/rounds_algo5.csv').read().split('\n')
for i,(x,y) in enumerate(zip(a,b)):
    xs=x.split(','); ys=y.split(',')
    if xs[:7]!=ys[:7] or xs[9:]!=ys[9:]:
        print("first difference at line", i); print(a[0]); print("launch:", a[max(1,i-2):i+3]); print("world :", b[max(1,i-2):i+3]); break
else:
    print("no difference in", min(len(a),len(b)), "lines")
PY
