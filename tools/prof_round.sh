# the end-of-round evidence run (GPU box): the GPU test suite, the default bench line, rocprofv3 kernel stats and the two
# PMC passes for the three modes that are benchmarked (K=48, K=60, per-barcode groups).  usage: bash tools/prof_round.sh r02
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; T=${1:-r02}
timeout 1500 python -m pytest $R/tests -m gpu -x -q > $O/t_$T.log 2>&1; grep -E "passed|failed|rror" $O/t_$T.log | tail -3
timeout 600 python $R/bench.py > $O/bench_$T.log 2>&1; tail -1 $O/bench_$T.log | cut -c1-400
for m in "k48:" "k60:--k 60" "grouped:--grouped"; do
  tag=${m%%:*}; fl=${m#*:}
  rm -rf $O/prof_${T}_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_$tag -- python $R/bench.py --reads 1e8 --steps 2 --warmup 1 --no-cpu-baseline $fl > $O/prof_${T}_$tag.log 2>&1
  grep metric $O/prof_${T}_$tag.log | cut -c1-160
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${T}_${tag}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${T}_${tag}_$c -- python $R/bench.py --reads 1e8 --steps 1 --warmup 1 --no-cpu-baseline $fl > $O/pmc_${T}_${tag}_$c.log 2>&1
  done
done
ls $O/prof_${T}_k48/*/ | head
