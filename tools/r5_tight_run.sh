#!/bin/bash
B="--steps 3 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust"
P="import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['config']; print(round(d['ms_per_step'],2), c['phase_ms_rank0']['count'], {k:c.get(k) for k in ('buckets_rank0','n_buckets','buckets_split_rank0','max_slots_rank0','retained_kmers_rank0')}, [k for k in c.keys()][:60])"
for m in 3 4 5; do echo -n "grouped min_freq $m: "; timeout 200 python bench.py $B --grouped --min-freq $m 2>/dev/null | python -c "$P"; done
