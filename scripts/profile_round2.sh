#!/bin/bash
# Runs on the GPU box (via gpurun): the rocprofv3 evidence of a round, into gpurun_out/<tag>/.
#   1. kernel-trace --stats of the driver's bench command (8 pools)            -> kernel_stats_8pools.csv
#   2. kernel-trace --stats of one C4 pool alone                               -> kernel_stats_one_pool.csv
#   3. PMC passes on the 8-pool bench: FETCH_SIZE, WRITE_SIZE (separate passes: the TCC slots of gfx950), one SQ pass
#   4. SQ pass on one pool alone (instruction mix / wait cycles of the placement kernels without neighbours)
#   5. kernel-trace --stats of the rebalancer sweep (C5)                       -> kernel_stats_rebalance.csv
# Counter passes carry no trace domain other than the kernel list (gpurun refuses pmc + sys/hip traces).
# Counter passes run the eight pools in LOCKSTEP launches (COOK_MATCH_SERVED=0): rocprofv3 serialises the dispatches it counts, and a
# served walker waits for launches that would then never start (it gives up after its time-out and lockstep launches finish the match
# anyway: slow, and not what one wants to count).  Same device functions either way: match_resolve2 is one round of the walker.
set -u
TAG=${1:-r02prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
B8="--steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --no-adjacent"
B1="--pools 1 --pending 125000 --running 50000 --offers 6250 --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras --no-adjacent"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
cd /tmp
kt() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/kt_$name
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -o kt --output-format csv -- "$@" > "$OUT/bench_under_rocprof_$name.json" 2> "$OUT/kt_$name.err"
  find /tmp/kt_$name -name '*kernel_stats*' -exec cp {} "$OUT/kernel_stats_$name.raw.csv" \;
  # the batched path's kernels are instances of ONE template (cook_multi<&kernel, ...>), printed mangled: the inner name in front
  python - "$OUT/kernel_stats_$name.raw.csv" "$OUT/kernel_stats_$name.csv" "$ROOT/scripts" <<'PY'
import csv, sys
sys.path.insert(0, sys.argv[3])
from kernel_names import short_kernel_name
rows = list(csv.DictReader(open(sys.argv[1])))
if rows:
    w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=["Kernel"] + list(rows[0].keys()), quoting=csv.QUOTE_ALL)
    w.writeheader()
    for r in rows:
        w.writerow({"Kernel": short_kernel_name(r["Name"]), **r})
PY
  rm -f "$OUT/kernel_stats_$name.raw.csv"
  head -8 "$OUT/kernel_stats_$name.csv" | cut -c1-150
}
pmc() {  # name, counters, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  COOK_MATCH_SERVED=0 timeout 400 rocprofv3 --pmc $ctr -d /tmp/pmc_$name -o p --output-format csv -- "$@" > /dev/null 2> "$OUT/pmc_$name.err"
  python "$ROOT/scripts/pmc_summary.py" /tmp/pmc_$name 24 > "$OUT/pmc_$name.txt"
  head -4 "$OUT/pmc_$name.txt" | cut -c1-400
}
for S in ${STEPS:-kt8 kt1 pmc8 sq1 ktrb}; do
  case $S in
    kt8) kt 8pools python "$ROOT/bench.py" $B8;;
    kt1) kt one_pool python "$ROOT/bench.py" $B1;;
    pmc8)
      pmc FETCH_SIZE_8pools FETCH_SIZE python "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-extras --no-adjacent --no-roofline
      pmc WRITE_SIZE_8pools WRITE_SIZE python "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-extras --no-adjacent --no-roofline
      pmc SQ_8pools "$SQ" python "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-extras --no-adjacent --no-roofline;;
    sq1) pmc SQ_one_pool "$SQ" python "$ROOT/bench.py" $B1 --no-roofline;;
    ktrb) kt rebalance python "$ROOT/scripts/bench_rebalance.py" --steps 2;;
  esac
done
ls "$OUT"
