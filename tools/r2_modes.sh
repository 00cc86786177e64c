cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for fl in "--k 60" "--grouped" "--sharded" "--reads 2e8" "--error-free" "--sorted-table"; do
  echo "== $fl"; timeout 600 python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 $fl 2>&1 | tail -1 | python3 -c "
import sys,json
l=sys.stdin.read().strip()
try:
    j=json.loads(l); print(j['value'], j['ms_per_step'], j['config'].get('phase_ms_rank0'), j['roofline']['launch_ms'])
except Exception as e: print('ERR', l[:300])
"
done
