"""Markdown table of per-kernel HBM traffic: python tools/traffic_table.py <kernel_stats.csv> <pmc FETCH csv> <pmc WRITE csv> [top]
(per launch: average duration from the kernel-stats run, FETCH_SIZE raw and WRITE_SIZE in GB from the two PMC passes)."""
import csv, sys, re
from collections import defaultdict
st, fc, wc = sys.argv[1:4]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 14


def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(anonymous namespace\)::", "", n); n = n.split("(")[0]
    return n[:60]


dur = {}
for r in csv.DictReader(open(st)):
    dur[short(r["Name"])] = float(r["AverageNs"]) / 1e6


def pmc(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"]); tot[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: tot[k] / n[k] * 1024 / 1e9 for k in tot}


f, w = pmc(fc, "FETCH_SIZE"), pmc(wc, "WRITE_SIZE")
rows = sorted(((dur[k], k) for k in dur if k in f or k in w), reverse=True)[:top]
print("| kernel | ms / launch | FETCH_SIZE GB (raw) | WRITE_SIZE GB | (2·fetch + write) / time |\n|---|---|---|---|---|")
for d, k in rows:
    ff, ww = f.get(k, 0.0), w.get(k, 0.0)
    print(f"| `{k}` | {d:.2f} | {ff:.1f} | {ww:.1f} | {(2 * ff + ww) / d:.1f} TB/s |")
