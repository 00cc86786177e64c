#!/bin/bash
# round 5's A/B driver, last form: the count kernel variants on the bench rows they were built for (run on the GPU box; every command under timeout).
# Earlier forms of this script -- limit / wait-bound / fill sweeps of the booked-slot kernel, cell / round / bucket sweeps of the bit filter,
# chunk-merge caps -- are quoted with their output in profiles/r05_count_booked_slots*.log, r05_count_screen_*.log, r05_graph_chunk_merge.log.
B="--steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust"
P="import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2), d['config']['phase_ms_rank0'])"
for x in "" "--k 60" "--grouped"; do
  echo -n "default switches $x: "; timeout 200 python bench.py $B $x 2>/dev/null | python -c "$P"
  echo -n "SNK_TUNING=count_tight=0,count_screen=0 $x: "; SNK_TUNING=count_tight=0,count_screen=0 timeout 200 python bench.py $B $x 2>/dev/null | python -c "$P"
done
for sw in "" "SNK_TUNING=count_screen_ng=0" "SNK_TUNING=count_tight=0"; do
  env $sw timeout 400 python tools/err_probe.py 1e8 e06,e15 2>&1 | grep -v amdgpu | grep "call 3" | sed "s/^/[$sw] /"
done
