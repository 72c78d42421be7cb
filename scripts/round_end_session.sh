#!/bin/bash
# The evidence of a round's shipped revision, in one GPU session: scripts/round_end_session.sh <tag>
#   kernel revision -> rocprofv3 passes (scripts/profile_round2.sh) -> pmc traffic json at that revision -> the driver's bench line (which then quotes
#   it) -> the GPU test suite -> a guarded fuzz sweep.  Everything lands in gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>_*).
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export GPU_MAX_HW_QUEUES=8
python scripts/kernel_rev.py > "$OUT/kernel_rev.txt"; cat "$OUT/kernel_rev.txt"
bash scripts/profile_round2.sh "$TAG" > "$OUT/profile.log" 2>&1; tail -5 "$OUT/profile.log"
cd "$ROOT"
python scripts/make_pmc_traffic.py "gpurun_out/$TAG" "$TAG" && cp "profiles/${TAG}_pmc_traffic.json" "$OUT/pmc_traffic.json"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?"; tail -3 "$OUT/bench.err"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("ms/step", d["ms_per_step"], "value", d["value"], "roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic")}, "cpu", d["cpu_baseline"]["value"], "x", d.get("speedup_vs_cpu_baseline"))
print("phase", d["phase_ms"], "boundary", {k: v for k, v in d["boundary"].items() if k.startswith("update_ms") and k != "update_ms_samples"})
print("extras", {k: (v.get("p50_cycle_ms") or v.get("p50_cycle_us") or v.get("ms_total")) for k, v in d["extra_configs"].items()})
PY
# the rank parts as pool batches against a thread and a stream per pool, at this revision (the cycle, and the reference's default K = 1000)
for v in "COOK_RANK_BATCHES=4" "COOK_RANK_BATCHES=1" "COOK_RANK_BATCH=0"; do
  for k in 0 1000; do
    env $v timeout 200 python bench.py --no-cpu-baseline --no-adjacent --no-extras --no-roofline --steps 30 --warmup 3 --considerable $k > "$OUT/ab.json" 2> "$OUT/ab.err"
    python - "$v" "$k" "$OUT/ab.json" <<'PY' | tee -a "$OUT/rank_batch.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    print(sys.argv[1], "K =", sys.argv[2] if sys.argv[2] != "0" else "all", "ms/cycle %.3f" % d["ms_per_step"], "phase", {k: round(v, 3) for k, v in d["phase_ms"].items()}, "parity", d.get("parity_checked"), "batch", d.get("rank_batch"))
except Exception as ex:
    print(sys.argv[1], sys.argv[2], "FAILED", ex)
PY
  done
done
rm -f "$OUT/ab.json" "$OUT/ab.err"
# the load of ONE rank of an N-GPU job (8 / 4 / 2 / 1 pools of the cluster), on this one GPU: what DESIGN.md 8's scaling prediction rests on
( echo "scripts/pools_per_gpu.sh at kernel revision $(cat "$OUT/kernel_rev.txt") (bench.py --as-rank-of N: the pools rank 0 of an N-GPU job holds; 6 steps)"; bash scripts/pools_per_gpu.sh ) > "$OUT/pools_per_gpu.txt" 2>&1
cat "$OUT/pools_per_gpu.txt"
( time timeout 900 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
( time timeout 600 python scripts/fuzz_sweep.py --guard --match 300 --rebalance 100 --multi 100 --seed 31337 ) > "$OUT/fuzz_gpu.txt" 2>&1; tail -5 "$OUT/fuzz_gpu.txt"
