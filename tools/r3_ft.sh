set -x
cd /root/repo
export PYTHONUNBUFFERED=1
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest > gpurun_out/r3_ft_bench_$name.log 2>&1
  tail -1 gpurun_out/r3_ft_bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', d['ms_per_step'], d['value'], d['config'].get('kernel_ms_rank0'), d['config'].get('phase_ms_rank0'))"
}
run sep SNK_TRIM_FUSED=0
run fused SNK_TRIM_FUSED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "trim" > gpurun_out/r3_ft_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_ft_tests.log
tail -3 gpurun_out/r3_ft_tests.log
