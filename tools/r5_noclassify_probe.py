import os, sys
sys.path.insert(0, '/root/repo')
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
e = Engine(0); sp = synth.synth_params(100_000_000, seed=0x5EED0001); rows, quals, bc = e.synth(sp)
for nc in ("0", "1", "0", "1"):
    os.environ["SNK_BL_NOCLASSIFY"] = nc
    for _ in range(3):
        r = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
    print("noclassify", nc, {k: round(v, 2) for k, v in r.graph_ms.items()}, "graph", round(r.phase_ms["graph"], 2), "boundary", r.n_boundary, "kmers", r.n_kmers, "unitigs", r.n_unitigs, flush=True)
