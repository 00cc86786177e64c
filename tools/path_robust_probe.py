"""f1 / f4 off the bench's operating point: read pathing, MarkDups and the barcode lists at 0.2 / 0.6 / 1.5 % sequencing errors (100 M reads).
Round 3: pathing 63.9 / 140.0 / 482.1 ms (the second pass takes 26 / 60 / 90 M reads; its cost is the K look-ups per error), dictionary
17.1 / 23.3 / 47.2 ms (7.7 k / 126 k / 675 k unitigs), MarkDups 11.5 ms throughout."""
import sys, time, math, torch
sys.path.insert(0, "/root/repo")
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = 100_000_000
e = Engine(0)
for ppm in (2000, 6000, 15000):
    sp = synth.synth_params(n, seed=0x5EED0042, sub_ppm=ppm)
    lam, term, cum = 150 * ppm / 1e6, math.exp(-150 * ppm / 1e6), 0.0
    for j in range(4):
        cum += term; sp.err_cdf[j] = min(0xFFFFFFFF, int(cum * 4294967296.0)); term *= lam / (j + 1)
    rows, quals, bc = e.synth(sp)
    res = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
    for rep in range(2):
        t0 = time.perf_counter()
        _, _, _, info = res.path_reads(rows, 150, quals, mark_dups=True, bc=bc, unitig_bcs=True, download=False)
        dt = (time.perf_counter() - t0) * 1e3
    print(f"errors {ppm/1e4:.2f} %: {res.n_unitigs} unitigs; dictionary {info['dict_ms']:.1f} ms, pathing {info['path_ms']:.1f} ms (second pass {info['n_slow']} reads), dups {info['dups']['ms'] if 'ms' in info['dups'] else info['dups']}, bcs {info['bcs_ms']:.1f}; whole call {dt:.0f} ms", flush=True)
    del rows, quals, bc, res
    e.release_cache()
