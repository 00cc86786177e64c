export TMPDIR=/tmp; cd /root/repo; O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigparity.py -x -q -k "hot or robust or golden_case or overflow or grouped or streamed or minbc or launch_shapes" 2>&1 | tail -5
timeout 900 python tools/r4_repeat_probe.py 1e7 2>&1 | tail -6
timeout 900 python tools/r4_repeat_probe.py 1e8 2>&1 | tail -6
