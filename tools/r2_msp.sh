cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
cd /tmp; timeout 120 python $R/tools/count_probe.py 1e8 0 2>&1 | grep -E "^dbg" | tail -2
