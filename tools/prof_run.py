"""One synthetic workload, a few steps -- the command rocprofv3 wraps."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
e = Engine(0)
sp = synth.synth_params(n, seed=0x5EED0001)
rows, quals, bc = e.synth(sp)
for i in range(steps):
    res = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
    print(i, res.n_instances, res.n_kmers, {k: round(v, 2) for k, v in res.phase_ms.items()}, "split", res.buckets_split, flush=True)
