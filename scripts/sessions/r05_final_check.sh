#!/bin/bash
# last look at the shipped tree on the GPU, as the driver will run it: smoke(), the GPU suite, the default bench line; and a larger guarded
# fuzz sweep whose multi-pool configurations alternate between rank parts per engine and rank parts as ONE pool batch (every pool its own K)
set -u
TAG=${1:-r05zi}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python scripts/kernel_rev.py | tee "$OUT/kernel_rev.txt"
( time python -c "import __graft_entry__ as g; g.smoke()" ) > "$OUT/smoke.log" 2>&1; tail -3 "$OUT/smoke.log"
( time timeout 900 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
( time timeout 600 python scripts/fuzz_sweep.py --guard --match 300 --rebalance 0 --multi 600 --seed 777001 ) > "$OUT/fuzz_gpu_multi600.txt" 2>&1; tail -4 "$OUT/fuzz_gpu_multi600.txt"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("ms/step", d["ms_per_step"], "value", d["value"], "parity", d["parity_checked"], "rank_batch", d["rank_batch"], "x cpu", d.get("speedup_vs_cpu_baseline"))
print("phase", d["phase_ms"])
print("extras", {k: (v.get("p50_cycle_ms") or v.get("p50_cycle_us") or v.get("ms_total")) for k, v in d["extra_configs"].items()})
PY
