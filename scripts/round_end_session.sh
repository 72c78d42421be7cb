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
( time timeout 900 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
( time timeout 600 python scripts/fuzz_sweep.py --guard --match 300 --rebalance 100 --multi 100 --seed 31337 ) > "$OUT/fuzz_gpu.txt" 2>&1; tail -5 "$OUT/fuzz_gpu.txt"
