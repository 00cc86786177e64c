"""one GPU, more reads than the one-pass partition's slots allow: bucket-range passes chosen by the library"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1]))
e = Engine(0); sp = synth.synth_params(n, seed=0x5EED0001); rows, quals, bc = e.synth(sp)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    r = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"reads {n}: passes {e.last_partition_passes()} wall {dt*1e3:.0f} ms = {r.n_instances / dt / 1e9:.1f} Gk-mers/s, buckets {r.n_buckets}, k-mers {r.n_kmers}, unitigs {r.n_unitigs}, "
          f"arena {r.scratch_bytes / 2**30:.0f} GB, free now {torch.cuda.mem_get_info()[0] / 2**30:.0f} GB", {k: round(v) for k, v in r.phase_ms.items()}, flush=True)
