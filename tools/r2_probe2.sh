# usage: bash tools/r2_probe2.sh "<variants>" ["<pytest -k expr>"]   -- quick parity subset + count-kernel timing of the main build and variants
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
python -m pytest $R/tests/test_gpu_parity.py -m gpu -x -q -k "${2:-golden or bucket_count or launch_shapes or synth_vs_oracle or k60_golden}" > $O/tq.log 2>&1; grep -E "passed|failed|rror" $O/tq.log | tail -3
python $R/tools/count_probe.py 1e8 0,1,2 > $O/cp_main.log 2>&1; grep dbg $O/cp_main.log
for v in $1; do SNK_LIB_PATH=$R/supernova_amd/variants/libsnk_$v.so python $R/tools/count_probe.py 1e8 0 > $O/cp_$v.log 2>&1; echo $v; grep -E "dbg|prof" $O/cp_$v.log | tail -2; done
