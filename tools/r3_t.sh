cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_asm_sn.py -x -q > gpurun_out/r3_t_tests.log 2>&1; tail -2 gpurun_out/r3_t_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ingest --no-next-rows --sharded > gpurun_out/r3_bench_t.log 2>&1
tail -1 gpurun_out/r3_bench_t.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['multi_gpu'])"
