export TMPDIR=/tmp; cd /root/repo; O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigparity.py -x -q -k "streamed or dense or bv_image or robust or golden_case or unsorted or grouped" 2>&1 | tail -3
timeout 900 python tools/r4_repeat_probe.py 1e7 2>&1 | tail -8
SNK_INGEST_TRACE=1 timeout 900 python bench.py --reads 1e7 --steps 1 --warmup 0 --no-cpu-baseline --no-next-rows --no-robust --ingest-files 256 --ingest-pairs 100000 2>&1 | grep -E "snk ingest|f3_ingest" | sed -e 's/.*"f3_ingest"/f3_ingest/' | cut -c1-900
SNK_INGEST_TRACE=1 timeout 900 python bench.py --reads 1e7 --steps 1 --warmup 0 --no-cpu-baseline --no-next-rows --no-robust --ingest-files 64 --ingest-pairs 400000 2>&1 | grep -E "snk ingest|f3_ingest" | sed -e 's/.*"f3_ingest"/f3_ingest/' | cut -c1-900
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', round(d['ms_per_step'],2), d['config']['phase_ms_rank0'], d['config']['graph_ms_rank0'], d['config']['step_includes'], json.dumps(d['roofline']['kernels'])[:900])"
