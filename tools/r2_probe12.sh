cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for tg in 600 750 900 1100; do
  echo "grouped target $tg"; SNK_TARGET_INST=$tg timeout 300 python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --grouped 2>&1 | grep metric | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config'].get('phase_ms_rank0'))"
done
for tg in 3000 3500 4000 4500; do
  echo "k60 target $tg"; SNK_TARGET_INST=$tg timeout 300 python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --k 60 2>&1 | grep metric | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config'].get('phase_ms_rank0'))"
done
