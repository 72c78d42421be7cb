for c in 1 2 4; do
COOK_MAX_RANK_CHAINS=$c timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --boundary --no-adjacent --no-check > gpurun_out/uc_$c.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/uc_$c.json')); b=d['boundary']; print($c, round(b['update_ms'],2), round(b['cycle_ms'],2), round(b['fetch_ms'],2))
PY
done
