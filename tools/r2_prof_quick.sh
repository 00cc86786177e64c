cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
rm -rf $O/prof_q
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q -- python $R/bench.py --reads 1e8 --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_q.log 2>&1
f=$(ls $O/prof_q/*/*kernel_stats.csv | head -1); head -32 $f | cut -d, -f1-4 | cut -c1-140
