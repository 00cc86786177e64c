# round 6: jobs that approach the device's memory on ONE GPU: slot capacity follows the memory pressure (5 -> 3 -> 1.5 sigma), then the 45 % budget, then passes
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],2), "ms", round(d["value"],2), "Gk-mers/s | phases", {k: round(v,1) for k,v in c["phase_ms_rank0"].items() if k in ("partition","count","graph")}, "| scratch GB", c.get("scratch_gb"), "overflow", c.get("overflow_supermers"), "reserved", c.get("arena_reserved_gb"))'
for n in 1e8 1.5e8 2e8 2.5e8 3e8; do
  echo -n "reads $n: "; timeout 600 python bench.py --reads $n --steps 3 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust --no-df-seam 2>/dev/null | python -c "$P"
done
