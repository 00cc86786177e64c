set -x
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 3000 python -m pytest tests -q -x -m gpu > gpurun_out/r3_full_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_full_tests.log
tail -4 gpurun_out/r3_full_tests.log
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-next-rows --no-ingest > gpurun_out/r3_bench_s.log 2>&1
tail -1 gpurun_out/r3_bench_s.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('single', d['ms_per_step'], d['value'], d['config'].get('phase_ms_rank0'), d['config'].get('graph_ms_rank0'))"
done
