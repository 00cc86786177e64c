# round 5: SQ instruction mix + LDS conflict counters of EVERY kernel of the step (bench workload, 1 timed step after 1 warm-up)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for pass in "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS"; do
  tag=${pass%%:*}; ctr=${pass#*:}
  rm -rf $O/pmc_r05_$tag
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_r05_$tag -- python $R/bench.py --reads 1e8 --steps 1 --warmup 1 --no-cpu-baseline --no-next-rows --no-ingest --no-robust > $O/pmc_r05_$tag.log 2>&1
  f=$(ls $O/pmc_r05_$tag/*/*counter_collection.csv | head -1)
  python - $f $O/r05_pmc_${tag}_by_kernel.csv <<'PY'
import csv, sys, collections
rd = csv.DictReader(open(sys.argv[1]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
for r in rd:
    k = r["Kernel_Name"][:100]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(r["Dispatch_Id"]); 
    if (k,key) not in seen: seen.add((k,key)); n[k]+=1
names = sorted({c for k in acc for c in acc[k]})
w = csv.writer(open(sys.argv[2], "w", newline=""))
w.writerow(["kernel", "dispatches"] + names)
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    w.writerow([k, n[k]] + [int(acc[k][c]) for c in names])
PY
  rm -rf $O/pmc_r05_$tag
  head -25 $O/r05_pmc_${tag}_by_kernel.csv | cut -c1-250
done
