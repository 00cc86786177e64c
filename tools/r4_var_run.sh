# round 4: the partition questions in one gpurun call -- the two-level building blocks (tools/probe/partition2.hip) and the bench
# line under tuning builds of the partition kernel (lane-paired record stores).  usage: bash tools/r4_var_run.sh [probe|bench|all]
export TMPDIR=/tmp; cd /root/repo
what=${1:-all}
if [ "$what" != "bench" ]; then
  timeout 600 tools/probe/_bin/partition2 2>&1 | tee gpurun_out/r04_partition2_probe.log
fi
if [ "$what" != "probe" ]; then
  B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust"
  ex() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d['ms_per_step'],2), d['config']['phase_ms_rank0'], d['roofline'].get('launch_ms'))" "$1"; }
  $B 2>/dev/null | ex base
  for v in $(ls supernova_amd/variants/libsnk_*.so 2>/dev/null); do
    n=$(basename $v .so)
    SNK_LIB_PATH=$PWD/$v timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or overflow or vs_oracle" 2>&1 | tail -1
    SNK_LIB_PATH=$PWD/$v $B 2>/dev/null | ex $n
  done
fi
