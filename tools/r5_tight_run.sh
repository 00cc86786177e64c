#!/bin/bash
# round 5: merged graph chunks for per-barcode groups (off by default there)
export SNK_CHUNK_MERGE=256
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 -k "grouped or group" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
B="--steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust"
P="import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), d['config']['phase_ms_rank0']['count'], d['config']['phase_ms_rank0']['graph'], d['config'].get('graph_ms_rank0'))"
for m in 0 256; do
export SNK_CHUNK_MERGE=$m
echo -n "merge=$m grouped: "; timeout 200 python bench.py $B --grouped 2>/dev/null | python -c "$P"
done
