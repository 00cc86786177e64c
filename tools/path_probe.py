"""f1 / f4 timing: read pathing on the device at bench size (dictionary build + pathing) and duplicate marking over the paths.
usage: python tools/path_probe.py [n_reads=1e8] [reps=2]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
e = Engine(0)
sp = synth.synth_params(n, seed=0x5EED0001)
rows, quals, bc = e.synth(sp, qstride=160)
for rep in range(reps):
    res = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    off, ne, edges, info = res.path_reads(rows, 150, quals, mark_dups=True, bc=bc, unitig_bcs=True)
    t1 = time.perf_counter()
    print(f"rep {rep}: {n} reads on {res.n_unitigs} unitigs ({res.n_kmers} k-mers): dictionary+tables {info['dict_ms']:.1f} ms, pathing {info['path_ms']:.1f} ms (second pass: {info['n_slow']} reads) "
          f"({n / info['path_ms'] / 1e3:.1f} M reads/s), HBV device part {info['hbv_device_ms']:.1f} ms, whole call incl. host flood and download {1e3 * (t1 - t0):.0f} ms; "
          f"paths: empty {int((ne == 0).sum())}, one edge {int((ne == 1).sum())}, more {int((ne > 1).sum())}, max {int(ne.max())}; "
          f"MarkDups {info['dups']['ms']:.1f} ms: {info['dups']['n_dup_pairs']} duplicate pairs, inter-barcode rate {info['dups']['interdup_rate']:.3f}; "
          f"unitig barcode lists: {len(info['unitig_bcs'][1])} entries", flush=True)
