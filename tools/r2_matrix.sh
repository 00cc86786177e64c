# usage: bash tools/r2_matrix.sh "<variants>" [modes]  -- count-kernel time of tuning builds at 1e8 reads
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for v in $1; do
  if [ $v = main ]; then unset SNK_LIB_PATH; else export SNK_LIB_PATH=$R/supernova_amd/variants/libsnk_$v.so; fi
  echo "== $v"; timeout 150 python $R/tools/count_probe.py 1e8 ${2:-0} 2>&1 | grep "^dbg"
done
