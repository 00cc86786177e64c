cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
cd /tmp
echo main; timeout 120 python $R/tools/count_probe.py 1e8 0,1,2 2>&1 | grep "^dbg"
echo alignbit; SNK_LIB_PATH=$R/supernova_amd/variants/libsnk_ab.so timeout 120 python $R/tools/count_probe.py 1e8 0 2>&1 | grep "^dbg"
