python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest 2>/dev/null | tail -1 > gpurun_out/r05_robust_line.json
python - <<'PY'
import json
p = json.loads(open('gpurun_out/r05_robust_line.json').read())
print('headline', round(p['ms_per_step'], 2), 'ms', round(p['value'], 2), 'Gk-mers/s', p['config']['phase_ms_rank0'])
for k, v in p['config']['robust'].items():
    if isinstance(v, dict): print(k, {kk: v[kk] for kk in ('ms', 'vs_headline_ms', 'first_call_ms', 'first_call_repartitioned', 'first_call_phases', 'calls_ms', 'phase_ms', 'buckets', 'buckets_split') if kk in v})
    else: print(k, str(v)[:100])
PY
