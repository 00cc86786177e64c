cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for t in 4000 6000 7000 8000; do echo "target $t"; SNK_TARGET_INST=$t timeout 120 python $R/tools/count_probe.py 1e8 0 2>&1 | grep "^dbg"; done
for p in 8 16 64; do echo "persist $p"; SNK_COUNT_PERSIST=$p timeout 120 python $R/tools/count_probe.py 1e8 0 2>&1 | grep "^dbg"; done
