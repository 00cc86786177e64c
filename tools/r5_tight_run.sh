#!/bin/bash
# round 5: grouped screen with 768-record batches and 14 rounds per lane (tuning build)
B="--steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust --grouped"
P="import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2), d['config']['phase_ms_rank0']['partition'], d['config']['phase_ms_rank0']['count'], d['config']['phase_ms_rank0']['graph'], d['config'].get('retained_kmers_rank0'), d['config'].get('unitigs_rank0'))"
echo -n "committed (512 records, 10 rounds, 5200): "; timeout 200 python bench.py $B 2>/dev/null | python -c "$P"
export SNK_LIB_PATH=$PWD/supernova_amd/variants/libsnk_b768.so
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --timeout 300 -k "grouped" 2>&1 | grep -E "passed|failed|Error|error" | tail -2
for t in 5200 6500 7500 8500; do echo -n "768 records target $t: "; SNK_TARGET_INST=$t timeout 200 python bench.py $B 2>/dev/null | python -c "$P"; done
