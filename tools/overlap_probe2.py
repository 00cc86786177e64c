"""The partition kernel once more on a second stream NEXT TO the count kernel of the same call (SNK_OVERLAP_PROBE, snk_pipeline.hip):
what does the hardware make of an atomics-bound and a VALU-bound kernel that are resident at the same time?
usage: python tools/overlap_probe2.py [reads]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
torch.cuda.set_device(0)
e = Engine(0)
sp = synth.synth_params(n, seed=0x5EED0001)
rows, quals, bc = e.synth(sp)
P = Params(K=48, sorted_table=False)
for _ in range(3):
    r = e.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=P)
print("baseline phases", r.phase_ms, "kernels", r.kernel_ms, flush=True)
for mode, dbg in ((2, 0), (1, 0), (5, 0), (2, 1), (1, 1), (2, 2), (1, 2), (2, 3), (1, 3)):
    e.set_option("overlap_probe", int(mode))
    e.set_option("overlap_probe_dbg", int(dbg))
    print(f"--- SNK_OVERLAP_PROBE={mode} relaunched kernel dbg={dbg} (0 whole kernel, 1 no record stores, 2 no slot atomics, 3 scan only)", flush=True)
    for _ in range(3):
        r = e.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=P)
        torch.cuda.synchronize()
e.set_option("overlap_probe", 0)
