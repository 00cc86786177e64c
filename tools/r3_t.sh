cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigparity.py -x -q > gpurun_out/r3_t_tests.log 2>&1; tail -2 gpurun_out/r3_t_tests.log
for i in 1 2; do
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ingest --no-next-rows > gpurun_out/r3_bench_t.log 2>&1
tail -1 gpurun_out/r3_bench_t.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['phase_ms_rank0'], d['config']['graph_ms_rank0'])"
done
