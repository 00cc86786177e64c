"""How the one-GPU step behaves off the bench's operating point: more sequencing errors (more distinct k-mers per bucket: hash-split
passes in the count kernel), more low-quality tails (the scan path of the fused trim), lower / higher coverage.
usage: python tools/robust_probe.py [n_reads]"""
import math, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
eng = Engine(0)


def cdf(lam):
    out, term, cum = [], math.exp(-lam), 0.0
    for j in range(4):
        cum += term
        v = cum * 4294967296.0
        out.append(0xFFFFFFFF if v >= 4294967295.0 else int(v))
        term *= lam / (j + 1)
    return out


import os
only = os.environ.get("ROBUST_ONLY")
cases = [("bench model (0.2 % errors, 5 % tails, 56x)", {}),
         ("50 % of the reads with a Q2 tail", dict(lowq_tail_ppm=500000)),
         ("0.6 % errors", dict(sub_ppm=6000)),
         ("1.5 % errors, 50 % tails", dict(sub_ppm=15000, lowq_tail_ppm=500000)),
         ("28x coverage", dict(genome_len=n * 150 // 28)),
         ("112x coverage", dict(genome_len=n * 150 // 112)),
         ("1000x coverage (a 15 Mb genome)", dict(genome_len=n * 150 // 1000)),
         ("100-base reads", dict(read_len=100, genome_len=n * 100 // 56)),
         ("250-base reads", dict(read_len=250, genome_len=n * 250 // 56, insert_min=500)),
         ("K=60, 0.6 % errors", dict(sub_ppm=6000, K=60))]
for name, ov in cases:
    if only and only not in name:
        continue
    K = ov.pop("K", 48)
    sp = synth.synth_params(n, seed=0x5EED0042, **ov)
    L = sp.read_len
    if "sub_ppm" in ov:
        for j, v in enumerate(cdf(L * ov["sub_ppm"] / 1e6)):
            sp.err_cdf[j] = v
    rows, quals, bc = eng.synth(sp)
    torch.cuda.synchronize()
    best = None
    eng2 = Engine(0)          # a fresh context per case: no hint from the case before
    for rep in range(3):
        t0 = time.perf_counter()
        r = eng2.count_graph(rows, L, quals=quals, bc=bc, params=Params(K=K, sorted_table=False))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if rep == 0: first = (dt, r.repartitioned, r.n_buckets)
        if best is None or dt < best[0]:
            best = (dt, dict(r.phase_ms), r.n_instances, r.n_kmers, r.n_unitigs, r.buckets_split, r.n_overflow, r.n_buckets, r.max_slots_used, rep, r.repartitioned)
    dt, ph, ni, nk, nu, bs, novf, nb, ms, brep, repart = best
    print(f"{name}: {dt:.1f} ms = {ni / dt / 1e6:.1f} Gk-mers/s | partition {ph['partition']:.1f} count {ph['count']:.1f} graph {ph['graph']:.1f} | "
          f"{nk} k-mers {nu} unitigs; buckets {nb}, split {bs}, overflow supermers {novf}, max slots {ms}; best of 3 = call {brep}", flush=True)
    print(f"    first call on a fresh context: {first[0]:.1f} ms (incl. allocation), repartitioned={first[1]}, buckets {first[2]}", flush=True)
    del rows, quals, bc, r
    eng2.close()
    eng.release_cache()
