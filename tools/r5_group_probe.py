"""per-barcode groups (config 5's shape) at 100 M reads: phases, buckets and splits per min_freq.  usage: python tools/r5_group_probe.py [n_reads] [min_freqs: 3,4]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
mfs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3,4").split(",")]
e = Engine(0)
sp = synth.synth_params(n, seed=0x5EED0001)
rows, quals, bc = e.synth(sp)
for mf in mfs:
    for rep in range(3):
        t0 = time.perf_counter()
        r = e.count_graph(rows, 150, quals=quals, bc=None, group=bc, params=Params(K=48, sorted_table=False, grouped=True, min_bc=0, min_freq=mf))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        print(f"min_freq {mf} call {rep}: {ms:.1f} ms | " + " ".join(f"{k} {v:.1f}" for k, v in r.phase_ms.items() if k in ("partition", "count", "graph")) +
              f" | buckets {r.n_buckets} split {r.buckets_split} kmers {r.n_kmers} unitigs {r.n_unitigs} max_slots {r.max_slots_used} limit {e.last_count_limit()}", flush=True)
