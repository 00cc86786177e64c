# round 6: what the partition's slot capacity (mean + k sigma of the occupancy model; msp_cap_pct scales it) costs and saves at 100 M reads:
# slot bytes, overflow supermers, partition / count / total ms.  usage: bash tools/r6_cap_sigma.sh
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],2), "ms | phases", {k: round(v,1) for k,v in c["phase_ms_rank0"].items() if k in ("partition","count","graph")}, "| scratch GB", c.get("scratch_gb"), "overflow", c.get("overflow_supermers"))'
for pct in 100 85 75 62 50; do
  echo -n "msp_cap_pct=$pct: "; SNK_TUNING=msp_cap_pct=$pct timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust --no-df-seam 2>/dev/null | python -c "$P"
done
