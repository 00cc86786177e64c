"""a14 on the device: time snk_dev_hbv (device part by HIP events, whole call by the wall clock) on unitig sets of
different sizes -- error-rich reads kept at min_freq=1 give millions of short unitigs.
usage: python tools/hbv_probe.py [n_reads ...]"""
import os
import sys
import time

import numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
e = Engine(0)
for arg in (sys.argv[1:] or ["1e6", "1e7"]):
    n = int(float(arg))
    for mf, mb in ((3, 2), (1, 0)):
        sp = synth.synth_params(n, seed=0x5EED0001)
        rows, quals, bc = e.synth(sp)
        res = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, min_freq=mf, min_bc=mb, sorted_table=False))
        ref = None
        for mode, env in (("host flood", {"SNK_HBV_DEV_MIN": "4000000000"}), ("device flood", {"SNK_HBV_DEV_MIN": "0"})):
            os.environ.update(env)
            for rep in range(2):
                t0 = time.perf_counter()
                h = res.hbv()
                dt = (time.perf_counter() - t0) * 1e3
            if ref is None:
                ref = h
            else:
                assert all(np.array_equal(ref[k], h[k]) for k in ("v_left", "v_right", "src", "is_rc", "fwd", "rev")), "floods differ"
            print(f"reads {n:>10} min_freq {mf} {mode:>12}: unitigs {res.n_unitigs:>9} bases {res.unitig_total_bases:>11}  HBV vertices "
                  f"{h['n_vertices']:>9} edges {h['n_edges']:>9}  device {h['device_ms']:7.2f} ms  call {dt:8.1f} ms", flush=True)
