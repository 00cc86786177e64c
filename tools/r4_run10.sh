export TMPDIR=/tmp; cd /root/repo; O=gpurun_out
SNK_ARENA_TRACE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --no-robust --no-next-rows 2>&1 | grep -E "snk arena|ms_per_step" | cut -c1-200 | tail -60
timeout 900 python -m pytest tests/test_gpu_rankshare.py -x -q 2>&1 | tail -30
