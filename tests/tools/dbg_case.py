"""Triage aid: replay one case of tests/tools/fuzz_parity.py and vary one factor at a time."""
import os, sys, threading, itertools
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tests" / "tools"))
import numpy as np, torch
import oracle_lib
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
from supernova_amd.sharded import ShardedEngine, SimWorld
sys.argv = [sys.argv[0]]
seed, want = 777, 52
rng = np.random.default_rng(seed)
import importlib.util
spec = importlib.util.spec_from_file_location("fz", ROOT / "tests" / "tools" / "fuzz_parity.py")
src = (ROOT / "tests" / "tools" / "fuzz_parity.py").read_text()
ns = {}
exec(src[src.index("def make_reads"):src.index("def same")], {"np": np}, ns)
make_reads = ns["make_reads"]
for case in range(want + 1):
    K = 48 if rng.random() < 0.7 else 60
    L = int(rng.choice([100, 150, 151, 250])); G = int(rng.choice([500, 3000, 20000, 120000])); cov = float(rng.choice([3, 8, 30, 60]))
    n = max(10, int(G * cov / L)); err = float(rng.choice([0.0, 0.002, 0.01])); nbc = int(rng.choice([1, 3, 40]))
    min_freq = int(rng.choice([1, 2, 3, 4])); min_bc = int(rng.choice([0, 1, 2])); nb = int(rng.choice([0, 0, 1, 5, 97, 4099])); use_bc = rng.random() < 0.8
    codes, quals, lens, bc = make_reads(rng, G, n, L, err, nbc, rng.random() < 0.6)
    rng.choice([2, 3, 5, 8])
print("case", want, dict(K=K, L=L, G=G, n=n, err=err, min_freq=min_freq, nb=nb))
dev = torch.device("cuda", 0)
rows = torch.from_numpy(synth.pack_rows(codes).view(np.int32)).to(dev); dq = torch.from_numpy(quals).to(dev); dbc = torch.from_numpy(bc).to(dev)
dl = torch.from_numpy(lens.view(np.int16)).to(dev)

def run(K, min_freq, W, nbt):
    gl = oracle_lib.good_lens(quals, lens, K=K)
    o = oracle_lib.OracleResult(codes, gl, bc, K=K, min_freq=min_freq, min_bc=0, hbv=False)
    world = SimWorld(W); bounds = [n * q // W for q in range(W + 1)]; out = [None] * W; errs = []
    def worker(q):
        try:
            torch.cuda.set_device(0); e = Engine(0); lo, hi = bounds[q], bounds[q + 1]
            sh = ShardedEngine(e, world.comm(q))
            rr = sh.count_graph(rows[lo:hi].contiguous(), L, quals=dq[lo:hi].contiguous(), bc=dbc[lo:hi].contiguous(), lens=dl[lo:hi].contiguous(),
                                params=Params(K=K, min_freq=min_freq, min_bc=0, n_buckets=nbt), read_index_base=lo)
            fr = rr.frags
            F, TB = int(fr.n_frags), int(fr.total_bases)
            nk = rr._dl(fr.nk, F * 4, np.uint32, (F,)); bo = rr._dl(fr.boff, F * 8, np.uint64, (F,)); bs = rr._dl(fr.bases, TB, np.uint8, (TB,))
            # per-fragment content as a multiset signature: (nk, bases) independent of the fragment order
            sig = sorted((int(nk[i]), bs[int(bo[i]):int(bo[i]) + int(nk[i]) + K - 1].tobytes()) for i in range(F))
            import hashlib
            hsh = hashlib.sha1(repr(sig).encode()).hexdigest()[:12]
            out[q] = (rr.unitigs() if q == 0 else None, hsh); e.close()
        except BaseException as ex:
            errs.append(ex); world.barrier_obj.abort()
    ts = [threading.Thread(target=worker, args=(q,)) for q in range(W)]; [t.start() for t in ts]; [t.join() for t in ts]
    if errs: return "ERR " + repr(errs[0])[:80]
    a, b = set(out[0][0]), set(o.unitigs)
    if a - b:
        comp = str.maketrans("ACGT", "TGCA")
        bad_h, bad_o = sorted(a - b, key=len), sorted(b - a, key=len)
        nrc = sum(1 for u in bad_h if u.translate(comp)[::-1] in b)
        print("   wrong unitigs that are the reverse complement of an oracle unitig:", nrc, "of", len(bad_h))
        for u in bad_h[:3]:
            cands = [v for v in bad_o if len(v) == len(u)]
            best = min(cands, key=lambda v: sum(x != y for x, y in zip(u, v))) if cands else None
            if best:
                d = [i for i, (x, y) in enumerate(zip(u, best)) if x != y]
                print("   len", len(u), "closest oracle unitig differs at", len(d), "positions", d[:12])
    return f"diff {len(a - b)} of {len(b)} unitigs, frags {[x[1] for x in out]}"
for K_, mf, W, nbt in [(60, 1, 4, 4), (60, 1, 5, 5), (60, 1, 6, 6), (60, 1, 7, 7), (60, 1, 5, 15), (60, 3, 5, 5), (60, 1, 10, 10), (48, 1, 5, 5)]:
    print(dict(K=K_, min_freq=mf, W=W, nb_total=nbt), "->", run(K_, mf, W, nbt), flush=True)
