# round 4, second GPU call: new tests, the full default bench line, the SQ instruction mix of the big kernels, the C1 CPU timing
export TMPDIR=/tmp; cd /root/repo; O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigparity.py -x -q -k "dense or bv_image or robust or golden_case" 2>&1 | tail -3
timeout 1500 python bench.py > $O/bench_r04a.log 2>&1; tail -1 $O/bench_r04a.log | cut -c1-6000
cd /tmp
rm -rf $OLDPWD/$O/pmc_r04_k48_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /root/repo/$O/pmc_r04_k48_sq -- python /root/repo/bench.py --reads 1e8 --steps 1 --warmup 1 --no-cpu-baseline --no-next-rows --no-ingest --no-robust > /root/repo/$O/pmc_r04_k48_sq.log 2>&1
cd /root/repo
python - $(ls $O/pmc_r04_k48_sq/*/*counter_collection.csv | head -1) $O/r04_pmc_instmix_1e8.csv <<'PY'
import csv, sys
rd = csv.DictReader(open(sys.argv[1]))
w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=rd.fieldnames)
w.writeheader()
for r in rd:
    if "snk_count_kernel" in r["Kernel_Name"] or "snk_msp_kernel" in r["Kernel_Name"]:
        w.writerow(r)
PY
rm -rf $O/pmc_r04_k48_sq
python tools/make_traffic.py --instmix 100000000_k48_single $O/r04_pmc_instmix_1e8.csv | cut -c1-1500
cp profiles/traffic.json $O/traffic_r04.json
timeout 1500 python bench.py --steps 2 --warmup 1 --no-next-rows --no-ingest --no-robust --cpu-sample-10m --cpu-threads 16 > $O/bench_r04_cpu10m.log 2>&1; tail -1 $O/bench_r04_cpu10m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('cpu_baseline')))"
