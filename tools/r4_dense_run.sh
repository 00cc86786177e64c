# round 4: the dense partition (SNK_TUNING=msp_dense=1) against the one-pass partition: parity subset, then the bench phases
export TMPDIR=/tmp; cd /root/repo
SNK_TUNING=msp_dense=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or overflow or vs_oracle or grouped or k60 or minbc" 2>&1 | tail -3
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust"
ex() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d['ms_per_step'],2), d['config']['phase_ms_rank0'], d['roofline'].get('launch_ms'), d['roofline']['real_traffic_GBs'].get('snk_msp_kernel_launch_ms'))" "$1"; }
$B 2>/dev/null | ex base
SNK_TUNING=msp_dense=1 $B 2>&1 | tail -1 | ex dense
SNK_TUNING=msp_dense=1 $B --k 60 2>&1 | tail -1 | ex dense_k60
$B --k 60 2>&1 | tail -1 | ex base_k60
