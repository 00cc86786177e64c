#!/bin/bash
# round 5: the count kernel's TIGHT variant switched on by the data (snk_pipeline.hip): suite + the bench rows
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -4
B="--steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust"
P="import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2), d['config']['phase_ms_rank0'])"
for x in "" "--grouped" "--k 60"; do echo -n "auto $x: "; timeout 200 python bench.py $B $x 2>/dev/null | python -c "$P"; done
timeout 300 python tools/err_probe.py 1e8 e06,e15 2>&1 | grep -v amdgpu | grep "call" | sed "s/^/auto /"
