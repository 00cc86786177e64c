cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
timeout 900 python -m pytest $R/tests/test_gpu_sharded.py $R/tests/test_gpu_bigparity.py -m gpu -q -x --durations=5 -k "${1:-sharded or simulated}" > $O/tsh.log 2>&1; tail -25 $O/tsh.log
