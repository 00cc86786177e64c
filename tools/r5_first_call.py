"""first call on new data of the same size (the context's sizing history is the other data's): bench reads -> 0.6 % -> 1.5 % -> bench, twice"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = 100_000_000
e = Engine(0)
if os.environ.get('R5_RESERVE_GB'): e.reserve(int(os.environ['R5_RESERVE_GB']) << 30); print('reserved GB', os.environ['R5_RESERVE_GB'], flush=True)
models = [("bench", {}), ("e06", dict(sub_ppm=6000)), ("e15", dict(sub_ppm=15000, lowq_tail_ppm=500000)), ("bench", {}), ("e06", dict(sub_ppm=6000)), ("e15", dict(sub_ppm=15000, lowq_tail_ppm=500000))]
for name, ov in models:
    sp = synth.synth_params(n, seed=0x5EED0042, **ov); rows, quals, bc = e.synth(sp); torch.cuda.synchronize()
    calls = []
    for _ in range(3):
        t0 = time.perf_counter(); r = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False)); torch.cuda.synchronize()
        calls.append((round((time.perf_counter() - t0) * 1e3), round(r.phase_ms["partition"]), round(r.phase_ms["count"]), round(r.phase_ms["graph"]), int(r.repartitioned)))
    print(os.environ.get("SNK_PILOT_EST", "1"), name, "calls (wall, partition, count, graph, repartitioned):", calls, "arena GB", round(r.scratch_bytes / 2**30), flush=True)
    del rows, quals, bc
