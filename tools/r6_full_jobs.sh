# round 6: jobs of 200-800 M reads on ONE GPU, held in the compact form (rows + good lengths + barcode ids); tools/r6_full_job.py
for n in 2e8 3e8 4e8 6e8 8e8; do
  timeout 900 python tools/r6_full_job.py $n 5e7 3 2>&1 | grep -E "^reads|\"call\": 2|Error|rror" | python -c '
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print("   call 3:", d["wall_s"], "s =", d["Gkmers_per_s"], "Gk-mers/s | instances", d["instances"], "retained", d["retained_kmers"], "unitigs", d["unitigs"], "| passes", d["passes"], "buckets", d["buckets"], "split", d["buckets_split"],
              "| count", d["phase_ms"]["count"], "graph", d["phase_ms"]["graph"], "ms | scratch GiB", d["scratch_gb"], "| fragments", d["n_fragments"], "| checks", d["min_count"] >= 3 and d["spectrum_adds_up"] and d["unitig_lengths_add_up"] and d["same_as_first_call"])
    else:
        print(l.strip()[:200])
'
done
echo "== the bench's own large jobs (reads held with their quality rows)"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d["config"]; print(round(d["ms_per_step"],2), "ms", round(d["value"],2), "Gk-mers/s | phases", {k: round(v,1) for k,v in c["phase_ms_rank0"].items() if k in ("partition","count","graph")}, "| scratch GB", c.get("scratch_gb"), "overflow", c.get("overflow_supermers"))'
for n in 1e8 2e8 3e8; do
  echo -n "reads $n: "; timeout 600 python bench.py --reads $n --steps 3 --warmup 2 --no-cpu-baseline --no-next-rows --no-ingest --no-robust --no-df-seam 2>/dev/null | python -c "$P"
done
