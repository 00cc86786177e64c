"""What the first call on a fresh context costs (5.2-5.9 s with the cached blocks): a new engine, the bench workload,
three calls; then the same with the cached blocks (SNK_ARENA_VMM=0 is read when the context first allocates).
usage: python tools/first_call_probe.py [n_reads=1e8] [1|0]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
# one mode per process (memory a process has freed is slow to get again -- ~30 ms per GB on this stack -- whoever asks for it)
for vmm in ((sys.argv[2],) if len(sys.argv) > 2 else ("1", "0")):
    e.set_option("arena_vmm", int(vmm))
    e = Engine(0)
    sp = synth.synth_params(n, seed=0x5EED0001)
    rows, quals, bc = e.synth(sp)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        r = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"SNK_ARENA_VMM={vmm}: calls {[round(t, 1) for t in ts]} ms, scratch {r.scratch_bytes / 1e9:.1f} GB", flush=True)
    del r, rows, quals, bc
    e.close()
    torch.cuda.empty_cache()
