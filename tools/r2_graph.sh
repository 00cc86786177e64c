cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -3
cd /tmp; timeout 300 python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep metric | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['config'].get('phase_ms_rank0'), j['config'].get('graph_ms_rank0'))"
