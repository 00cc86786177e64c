cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for t in 512 640 896 1024; do
  for tg in 5000 6000; do
    echo "threads $t target $tg"; SNK_TARGET_INST=$tg SNK_LIB_PATH=$R/supernova_amd/variants/libsnk_t$t.so timeout 120 python $R/tools/count_probe.py 1e8 0 2>&1 | grep "^dbg"
  done
done
