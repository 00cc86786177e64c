# the end-of-round evidence run (GPU box): the GPU test suite, the default bench line, rocprofv3 kernel stats and the two PMC passes
# for the three modes that are benchmarked (K=48, K=60, per-barcode groups), the sharded path on a one-rank RCCL communicator,
# read pathing.  Every command under `timeout`.  usage: bash tools/prof_round.sh r03 [notests]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; T=${1:-r03}
# raw rocprofv3 output (hundreds of MB of counter rows) stays on the box: gpurun copies back at most 64 MiB of gpurun_out/
W=/tmp/snk_prof_$T; mkdir -p $W $O
if [ "$2" != "notests" ]; then
  timeout 2400 python -m pytest $R/tests -m gpu -x -q > $O/t_$T.log 2>&1; grep -E "passed|failed|rror" $O/t_$T.log | tail -3
fi
timeout 1500 python $R/bench.py > $O/bench_$T.log 2>&1; tail -1 $O/bench_$T.log | cut -c1-300
for m in "k48:" "k60:--k 60" "grouped:--grouped" "sharded:--sharded --no-verify"; do
  tag=${m%%:*}; fl=${m#*:}
  rm -rf $W/prof_${T}_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/prof_${T}_$tag -- python $R/bench.py --reads 1e8 --steps 3 --warmup 1 --no-cpu-baseline --no-next-rows --no-ingest --no-robust --no-df-seam $fl > $O/prof_${T}_$tag.log 2>&1
  grep metric $O/prof_${T}_$tag.log | cut -c1-160
  cp $(ls $W/prof_${T}_$tag/*/*kernel_stats.csv | head -1) $O/${T}_bench_1e8_${tag}_kernel_stats.csv
  if [ "$tag" = "k48" ]; then
    # instruction mix of the two big kernels on the bench workload (SQ block: 8 slots per pass): the count kernel's VALU-issue roofline
    rm -rf $W/pmc_${T}_${tag}_sq
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $W/pmc_${T}_${tag}_sq -- python $R/bench.py --reads 1e8 --steps 1 --warmup 1 --no-cpu-baseline --no-next-rows --no-ingest --no-robust --no-df-seam $fl > $O/pmc_${T}_${tag}_sq.log 2>&1
    python - $(ls $W/pmc_${T}_${tag}_sq/*/*counter_collection.csv | head -1) $O/${T}_pmc_instmix_1e8.csv <<'PY'
import csv, sys
# keep the rows of the two big kernels only (the full collection is tens of MB)
rd = csv.DictReader(open(sys.argv[1]))
w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=rd.fieldnames)
w.writeheader()
for r in rd:
    if "snk_count_kernel" in r["Kernel_Name"] or "snk_msp_kernel" in r["Kernel_Name"]:
        w.writerow(r)
PY
  fi
  if [ "$tag" != "sharded" ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $W/pmc_${T}_${tag}_$c
      timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $W/pmc_${T}_${tag}_$c -- python $R/bench.py --reads 1e8 --steps 1 --warmup 1 --no-cpu-baseline --no-next-rows --no-ingest --no-robust --no-df-seam $fl > $O/pmc_${T}_${tag}_$c.log 2>&1
      cp $(ls $W/pmc_${T}_${tag}_$c/*/*counter_collection.csv | head -1) $O/${T}_bench_1e8_${tag}_pmc_$c.csv
    done
  fi
done
rm -rf $W/prof_${T}_path
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/prof_${T}_path -- python $R/tools/path_probe.py 1e8 2 > $O/prof_${T}_path.log 2>&1
cp $(ls $W/prof_${T}_path/*/*kernel_stats.csv | head -1) $O/${T}_path_1e8_kernel_stats.csv; tail -1 $O/prof_${T}_path.log | cut -c1-200
for x in "--k 60" "--grouped" "--error-free" "--reads 2e8" "--sorted-table" "--sharded" "--minimiser 20" "--sharded --minimiser 20"; do
  timeout 600 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust --no-df-seam $x 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$x', round(d['ms_per_step'],2), round(d['value'],2))"
done > $O/bench_modes_$T.log 2>&1
cat $O/bench_modes_$T.log
