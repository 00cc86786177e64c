cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
SNK_LIB_PATH=$R/supernova_amd/variants/libsnk_prof.so timeout 120 python $R/tools/count_probe.py 1e8 0 2>&1 | grep -E "prof|^dbg" | tail -2
for p in 8 16 64; do echo "persist $p"; SNK_COUNT_PERSIST=$p timeout 120 python $R/tools/count_probe.py 1e8 0 2>&1 | grep "^dbg"; done
for t in 4000 6000; do echo "target $t"; SNK_TARGET_INST=$t timeout 120 python $R/tools/count_probe.py 1e8 0 2>&1 | grep "^dbg"; done
