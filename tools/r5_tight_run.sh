#!/bin/bash
# round 5: chunks per workgroup of the small-chunk prune (SNK_BL_CPW) now that merged-away chunks are empty
B="--steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust"
P="import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), d['config']['phase_ms_rank0']['graph'], d['config'].get('graph_ms_rank0'))"
for m in 1 2 4 8; do
export SNK_BL_CPW=$m
echo -n "cpw=$m: "; timeout 200 python bench.py $B 2>/dev/null | python -c "$P"
timeout 300 python tools/err_probe.py 1e8 e15 2>&1 | grep -v amdgpu | grep "call 3" | sed "s/^/cpw=$m /"
done
