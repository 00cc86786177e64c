# quick parity subset + count-kernel timing of the in-tree build over bucket sizes (SNK_TARGET_INST) + instruction mix
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
timeout 300 python -m pytest $R/tests/test_gpu_parity.py -m gpu -x -q -k "${2:-golden or bucket_count or launch_shapes or synth_vs_oracle or k60_golden or grouped}" > $O/tq.log 2>&1; grep -E "passed|failed|rror" $O/tq.log | tail -3
for t in $1; do echo "target $t"; SNK_TARGET_INST=$t timeout 120 python $R/tools/count_probe.py 1e8 0 2>&1 | grep "dbg"; done
timeout 120 python $R/tools/count_probe.py 1e8 1,2 2>&1 | grep "^dbg"
bash $R/tools/r2_pmc.sh main 2>&1 | tail -2
