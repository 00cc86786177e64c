"""a14 at scale, one size per process (a size that misbehaves must not take the others with it): unitigs of n_reads error-carrying
reads kept at min_freq = 1 -- the connected bulk of the genome graph plus hundreds of thousands of small components -- through the
host flood and through the device flood (components on the device, SNK_HBV_STRICT: a bounded loop that runs out is an error here).
usage: python tools/hbv_scale_probe.py n_reads [min_freq=1] [big_limit|-] [grouped]     (grouped: per-barcode graphs -- every component is small)"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1]))
mf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
grouped = len(sys.argv) > 4 and sys.argv[4] == "grouped"
e = Engine(0)
if len(sys.argv) > 3 and sys.argv[3] != "-":
    e.set_option("hbv_big", int(sys.argv[3]))
sp = synth.synth_params(n, seed=0x5EED0001)
rows, quals, bc = e.synth(sp)
if grouped:
    res = e.count_graph(rows, 150, quals=quals, bc=None, group=bc, params=Params(K=48, min_freq=mf, min_bc=0, sorted_table=False, grouped=True))
else:
    res = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, min_freq=mf, min_bc=0 if mf == 1 else 2, sorted_table=False))
print(f"reads {n} min_freq {mf}{' grouped' if grouped else ''}: unitigs {res.n_unitigs} bases {res.unitig_total_bases}", flush=True)
ref = None
for mode, opts in (("host flood", {"hbv_dev_min": 4000000000}), ("device flood", {"hbv_dev_min": 0, "hbv_strict": 1})):
    for k_, v_ in opts.items():
        e.set_option(k_, v_)
    for rep in range(2):
        t0 = time.perf_counter()
        h = res.hbv()
        dt = (time.perf_counter() - t0) * 1e3
        print(f"  {mode:>12} rep {rep}: vertices {h['n_vertices']} edges {h['n_edges']}  device part {h['device_ms']:.2f} ms  call {dt:.1f} ms", flush=True)
    if ref is None:
        ref = h
    else:
        same = all(np.array_equal(ref[k], h[k]) for k in ("v_left", "v_right", "src", "is_rc", "fwd", "rev"))
        print("  floods equal:", same, flush=True)
        assert same
