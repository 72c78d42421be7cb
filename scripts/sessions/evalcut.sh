#!/bin/bash
# timing study: eval scan phase per wave with parts of the per-offer work cut out (results are wrong in the cut builds)
cd ${GRAFT_REPO_ROOT:-.}
for l in ${EC_LIBS:-trace tc1 tc2 tc3}; do
  COOK_LIB=$PWD/cook_amd/libcookmatch_$l.so python bench.py --pools 1 --pending 125000 --running 50000 --offers 6250 --steps 1 --warmup 0 --no-cpu-baseline --no-adjacent --no-check --no-extras > /dev/null 2> /tmp/ec_$l.err
  python - "$l" <<'PY'
import re, sys
ph=[]
for line in open('/tmp/ec_%s.err' % sys.argv[1]):
    if line.startswith('EVALTRACE launch'): ph.append([])
    m=re.match(r"EVALPHASE blk (\d+) wave (\d) setup ([\d.-]+) scan ([\d.-]+) merge ([\d.-]+)", line)
    if m and ph: ph[-1].append(tuple(float(x) for x in m.groups()))
for k, r in enumerate(ph):
    if not r: continue
    v=sorted(x[3] for x in r)
    print(sys.argv[1], "launch#", k, "scan us: p10 %.2f p50 %.2f p90 %.2f (n=%d)" % (v[len(v)//10], v[len(v)//2], v[len(v)*9//10], len(v)))
PY
done
