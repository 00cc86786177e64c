cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
timeout 1500 python -m pytest $R/tests/test_gpu_rankshare.py -m gpu -q -x --durations=0 > $O/tbig.log 2>&1; tail -40 $O/tbig.log
