for c in 4 8 6; do
COOK_MAX_RANK_CHAINS=$c timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-adjacent --no-check > gpurun_out/rc_$c.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/rc_$c.json')); print($c, round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['phase_ms'].items()})
PY
done
