cd /root/repo
for v in 4000 4500 5000 5500 6000; do
SNK_TARGET_INST=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest > gpurun_out/r3_bench_ti.log 2>&1
tail -1 gpurun_out/r3_bench_ti.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('target', $v, round(d['ms_per_step'],2), d['config'].get('phase_ms_rank0'))"
done
