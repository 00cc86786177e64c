cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
timeout 300 python $R/tools/path_probe.py 1e8 2 2>&1 | tail -1
rm -rf $O/pmc_path
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/pmc_path -- python $R/tools/path_probe.py 1e7 1 > $O/pmc_path.log 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(float)
for f in glob.glob("$O/pmc_path/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'path_kernel' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
print({k:round(v/1e6,1) for k,v in agg.items()}, "(M, 1e7 reads)")
PY
