export TMPDIR=/tmp; cd /root/repo
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest"
ex() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d['ms_per_step'],2), d['config']['phase_ms_rank0'], d['roofline']['launch_ms'])" "$1"; }
$B 2>/dev/null | ex base
SNK_LIB_PATH=$PWD/supernova_amd/variants/libsnk_nt.so $B 2>/dev/null | ex nt_stores
SNK_TUNING=msp_cap_pct=160 $B 2>/dev/null | ex cap160
SNK_TUNING=msp_cap_pct=130 $B 2>/dev/null | ex cap130
