export TMPDIR=/tmp; cd /root/repo; O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigparity.py tests/test_gpu_ingest.py -x -q -k "hot or robust or golden_case or overflow or ingest or streamed" 2>&1 | tail -5
timeout 600 python tools/r4_repeat_probe.py 1e8 2>&1 | tail -6
SNK_INGEST_TRACE=1 timeout 900 python bench.py --reads 1e7 --steps 1 --warmup 0 --no-cpu-baseline --no-next-rows --no-robust --ingest-files 64 --ingest-pairs 400000 2>&1 | grep -E "snk ingest|f3_ingest" | sed -e 's/.*"f3_ingest"/f3_ingest/' -e 's/"roofline".*//' | cut -c1-1500
SNK_INGEST_TRACE=1 timeout 900 python bench.py --reads 1e7 --steps 1 --warmup 0 --no-cpu-baseline --no-next-rows --no-robust --ingest-files 128 --ingest-pairs 200000 2>&1 | grep -E "snk ingest|f3_ingest" | sed -e 's/.*"f3_ingest"/f3_ingest/' -e 's/"roofline".*//' | cut -c1-1500
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', round(d['ms_per_step'],2), d['config']['phase_ms_rank0']); print(json.dumps(d['config']['robust'],indent=0)[:3000])"
