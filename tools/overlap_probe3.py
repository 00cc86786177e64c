"""A LEAN emitter (no LDS, few registers: slot reservation + 32-byte record store per thread) on a second stream NEXT TO the count kernel
(SNK_OVERLAP_PROBE, snk_pipeline.hip / snk_stages.hip probe_lean_emit_kernel): do an atomics-bound and a VALU-bound kernel share the CUs
when the second one fits beside the first one's workgroups?   usage: FILES="snk_stages snk_pipeline" tools/build_variant.sh probes -DSNK_PROBES; SNK_LIB_PATH=supernova_amd/variants/libsnk_probes.so python tools/overlap_probe3.py [reads]"""
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
VARIANTS = ((2, 17), (1, 17), (2, 18), (1, 18), (2, 20), (1, 20)) if len(sys.argv) > 2 and sys.argv[2] == 'throttled' else ((2, 20), (1, 20), (2, 24), (1, 24), (5, 24), (2, 32), (1, 32), (2, 48), (1, 48))
for mode, dbg in VARIANTS:
    e.set_option("overlap_probe", int(mode))
    e.set_option("overlap_probe_dbg", int(dbg))
    print(f"--- SNK_OVERLAP_PROBE={mode} (2 alone, 1 next to the count kernel, 5 = 1 with a high-priority stream) lean emitter dbg={dbg}", flush=True)
    for _ in range(2):
        r = e.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=P)
        torch.cuda.synchronize()
e.set_option("overlap_probe", 0)
