# usage: bash tools/r2_pmc.sh "<variants (main = the in-tree build)>" [reads]   -- instruction mix of the count kernel per build
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for v in $1; do
  if [ $v = main ]; then unset SNK_LIB_PATH; else export SNK_LIB_PATH=$R/supernova_amd/variants/libsnk_$v.so; fi
  rm -rf $O/pmc_mix_$v
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/pmc_mix_$v -- timeout 120 python $R/tools/count_probe.py ${2:-1e7} 0 > $O/pmc_mix_$v.log 2>&1
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("$O/pmc_mix_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'snk_count_kernel' in k or 'snk_msp_kernel' in k:
            k=k.split('(')[1 if k.startswith('void (') else 0][:40] if False else ('count' if 'snk_count' in k else 'msp')
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k,v in agg.items():
    print("$v",k,{c:round(x/n[(k,c)]/1e6,1) for c,x in v.items()},"(M per launch)")
PY
done
