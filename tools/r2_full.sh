cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
cd $R
timeout 1700 python -m pytest tests -m gpu -q -x --durations=8 > $O/t_full.log 2>&1; tail -15 $O/t_full.log
timeout 600 python bench.py > $O/bench_full.log 2>&1; tail -1 $O/bench_full.log | cut -c1-1500
