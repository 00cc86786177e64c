export TMPDIR=/tmp; cd /root/repo
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust"
ex() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d['ms_per_step'],2), d['config']['phase_ms_rank0'], d['config']['graph_ms_rank0'])" "$1"; }
$B 2>/dev/null | ex base
SNK_LIB_PATH=$PWD/supernova_amd/variants/libsnk_fragnobytes.so $B 2>/dev/null | ex fragnobytes
timeout 900 python -m pytest tests/test_gpu_seam.py tests/test_gpu_sharded.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
