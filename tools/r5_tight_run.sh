#!/bin/bash
# round 5: grouped screen as committed (1024-slot table, 512-record batches, target 5200): suite + lines
timeout 2400 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/r5_suite.log 2>&1; grep -E "passed|failed|error" gpurun_out/r5_suite.log | tail -3
B="--steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust --grouped"
P="import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2), d['config']['phase_ms_rank0']['partition'], d['config']['phase_ms_rank0']['count'], d['config']['phase_ms_rank0']['graph'], d['config'].get('retained_kmers_rank0'), d['config'].get('unitigs_rank0'))"
echo -n "grouped: "; timeout 200 python bench.py $B 2>/dev/null | python -c "$P"
echo -n "grouped min_freq 4: "; timeout 200 python bench.py $B --min-freq 4 2>/dev/null | python -c "$P"
echo -n "grouped 2e8: "; timeout 300 python bench.py $B --reads 2e8 2>/dev/null | python -c "$P"
timeout 600 python tools/r5_group_probe.py 1e8 3 2>&1 | grep -v amdgpu | tail -3
