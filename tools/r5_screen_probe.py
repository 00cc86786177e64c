"""the count kernel's singleton screen on the bench's reads and on the error-rich models: SNK_COUNT_SCREEN=0 (off) / 1 (by the data's ratio) / 2 (always)"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
models = {"headline": {}, "e06": dict(sub_ppm=6000), "e15": dict(sub_ppm=15000, lowq_tail_ppm=500000)}
e = Engine(0)
for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else list(models)):
    sp = synth.synth_params(n, seed=0x5EED0042, **models[name]); rows, quals, bc = e.synth(sp)
    ref = None
    for mode in ("0", "1", "2"):
        os.environ["SNK_COUNT_SCREEN"] = mode
        calls = []
        for _ in range(3):
            r = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
            calls.append(round(r.phase_ms["total"], 1))
        sig = (r.n_kmers, r.n_unitigs, r.unitig_total_bases)
        ref = ref or sig
        print(name, "screen", mode, "calls", calls, "buckets", r.n_buckets, "split", r.buckets_split, {k: round(v, 1) for k, v in r.phase_ms.items() if k in ("partition", "count", "graph")},
              "kernel", round(r.kernel_ms["count"], 1), "same result:", sig == ref, flush=True)
    del rows, quals, bc
