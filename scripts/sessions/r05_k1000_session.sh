#!/bin/bash
# where the K = 1000 cycle's time goes (the reference's default cap, config.clj:113): kernel stats and the walkers' / servers' own accounts
set -u
TAG=${1:-r05zj}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
python scripts/kernel_rev.py | tee "$OUT/kernel_rev.txt"
B="--steps 40 --warmup 5 --considerable 1000 --no-cpu-baseline --no-check --no-extras --no-adjacent --no-roofline"
( cd /tmp && rm -rf /tmp/kt_k && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_k -o kt --output-format csv -- python "$ROOT/bench.py" $B > "$OUT/bench_under_rocprof_k1000.json" 2> "$OUT/kt_k1000.err"; find /tmp/kt_k -name '*kernel_stats*' -exec cp {} "$OUT/kernel_stats_k1000.raw.csv" \; )
python - "$OUT/kernel_stats_k1000.raw.csv" "$OUT/kernel_stats_k1000.csv" "$ROOT/scripts" <<'PY'
import csv, sys
sys.path.insert(0, sys.argv[3])
from kernel_names import short_kernel_name
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=["Kernel"] + list(rows[0].keys()), quoting=csv.QUOTE_ALL)
w.writeheader()
for r in rows:
    w.writerow({"Kernel": short_kernel_name(r["Name"]), **r})
for r in rows[:14]:
    print("%-36s calls/cycle %6.1f avg %8.1f us  per cycle %8.1f us" % (short_kernel_name(r["Name"])[:36], int(r["Calls"]) / 45.0, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 45e3))
PY
rm -f "$OUT/kernel_stats_k1000.raw.csv"
COOK_SERVE_TRACE=1 timeout 200 python bench.py --steps 3 --warmup 2 --considerable 1000 --no-cpu-baseline --no-check --no-extras --no-adjacent --no-roofline > "$OUT/k1000_line.json" 2> "$OUT/k1000_serve_trace.err"
grep SERVETRACE "$OUT/k1000_serve_trace.err" | tail -11 > "$OUT/k1000_serve_trace.txt"; cat "$OUT/k1000_serve_trace.txt"
python - <<PY
import json
d = json.loads(open("$OUT/k1000_line.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["phase_ms"], d["last_cycle"]["placement_stats_pool0"])
PY
