"""Phase timings of the device path at increasing synthetic sizes (GPU box)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params

sizes = [int(float(x)) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1_000_000, 10_000_000]
ef = len(sys.argv) > 2 and sys.argv[2] == "clean"
e = Engine(0)
for n in sizes:
    sp = synth.synth_params(n, seed=0x5EED0001, error_free=ef)
    t0 = time.time()
    rows, quals, bc = e.synth(sp)
    torch.cuda.synchronize()
    t1 = time.time()
    for rep in range(2):
        torch.cuda.synchronize(); t2 = time.time()
        res = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
        torch.cuda.synchronize(); t3 = time.time()
        print(f"n={n} rep={rep} gen={t1-t0:.2f}s wall={t3-t2:.3f}s inst={res.n_instances} Gk/s={res.n_instances/(t3-t2)/1e9:.3f} "
              f"super={res.n_supermers} NB={res.n_buckets} kmers={res.n_kmers} unitigs={res.n_unitigs} split={res.buckets_split} "
              f"maxslots={res.max_slots_used} rounds={res.rank_rounds} scratchGB={res.scratch_bytes/1e9:.1f}", flush=True)
        print("   phases(ms):", {k: round(v, 2) for k, v in res.phase_ms.items()}, flush=True)
    del rows, quals, bc
