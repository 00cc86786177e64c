"""Per-step kernel times from a rocprofv3 --kernel-trace --stats csv: usage python tools/kstat.py <kernel_stats.csv> <steps incl. warmup and setup>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = 0
for r in rows:
    n = r["Name"]
    short = n.split("(")[0] if "anonymous" not in n else n.split("::", 1)[1].split("(")[0]
    per = float(r["TotalDurationNs"]) / 1e6 / steps
    if "synth" in short:
        continue
    tot += per
    if per > 0.04:
        print(f"{short[-70:]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e6:8.3f} ms  per-step {per:7.3f}")
print(f"sum per step {tot:.2f} ms")
