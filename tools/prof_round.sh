cd /tmp && export TMPDIR=/tmp
R=/root/repo
python -m pytest $R/tests -m gpu -x -q > $R/gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" $R/gpurun_out/t.log | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01k -- python $R/bench.py --reads 1e8 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r01k.log 2>&1
grep metric $R/gpurun_out/prof_r01k.log | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmck_$c -- python $R/bench.py --reads 1e8 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmck_$c.log 2>&1; done
ls $R/gpurun_out/prof_r01k/*/ | head
