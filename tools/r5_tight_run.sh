#!/bin/bash
# round 5: the TIGHT count kernel with larger buckets on the bench's own (clean) reads
B="--steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust"
P="import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2), d['config']['phase_ms_rank0'])"
echo -n "default: "; timeout 200 python bench.py $B 2>/dev/null | python -c "$P"
for t in 5000 6000 7000 8000 9000; do echo -n "tight=1920 target $t: "; SNK_COUNT_TIGHT=1920 SNK_TARGET_INST=$t timeout 200 python bench.py $B 2>/dev/null | python -c "$P"; done
for t in 4500 5500; do echo -n "k60 tight=1920 target $t: "; SNK_COUNT_TIGHT=1920 SNK_TARGET_INST=$t timeout 200 python bench.py $B --k 60 2>/dev/null | python -c "$P"; done
