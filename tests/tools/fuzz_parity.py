"""Randomised parity sweep: the HIP path (single-GPU, both graph stages, and W simulated ranks) against the C oracle
(oracle/snk_oracle.c) over random genomes / read sets / parameters.  Test infrastructure, like tests/: it may use the oracle.
usage: python tests/tools/fuzz_parity.py [n_cases] [seed]"""
import os, sys, threading
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import oracle_lib
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
from supernova_amd.sharded import ShardedEngine, SimWorld

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1      # replay one case of a sweep (same random stream)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
dev = torch.device("cuda", 0)
eng = Engine(0)


def make_reads(rng, G, n, L, err, nbc, repeat):
    g = rng.integers(0, 4, G, dtype=np.uint8)
    if repeat:                                    # planted repeats, a tandem run, a palindrome
        rep = g[100:100 + int(rng.integers(60, 400))].copy()
        for _ in range(int(rng.integers(1, 5))):
            p = int(rng.integers(0, G - len(rep))); g[p:p + len(rep)] = rep
        unit = rng.integers(0, 4, int(rng.integers(1, 9)), dtype=np.uint8)
        p = int(rng.integers(0, G - 200)); g[p:p + 160] = np.resize(unit, 160)
        x = rng.integers(0, 4, 30, dtype=np.uint8); p = int(rng.integers(0, G - 60)); g[p:p + 60] = np.concatenate([x, (3 - x[::-1])])
    codes = np.zeros((n, L), dtype=np.uint8); quals = np.full((n, L), 30, dtype=np.uint8); lens = np.full(n, L, dtype=np.uint16)
    for i in range(n):
        ln = L if rng.random() < 0.8 else int(rng.integers(20, L + 1))
        ln = min(ln, G)
        s = int(rng.integers(0, G - ln + 1)); r = g[s:s + ln].copy()
        if rng.random() < 0.5: r = (3 - r[::-1]).astype(np.uint8)
        e = rng.random(ln) < err; r[e] = (r[e] + 1 + rng.integers(0, 3, int(e.sum()))) & 3; quals[i, :ln][e] = 12
        if rng.random() < 0.1: quals[i, int(rng.integers(0, ln)):ln] = 2
        codes[i, :ln] = r; lens[i] = ln
    bc = rng.integers(0, nbc + 1, n).astype(np.int32)
    return codes, quals, lens, bc


def same(res_keys, res_counts, res_ctx, res_unitigs, o):
    return (np.array_equal(res_keys, o.keys) and np.array_equal(res_counts, o.counts) and np.array_equal(res_ctx, o.ctx)
            and res_unitigs == o.unitigs)


bad = 0
for case in range(n_cases):
    K = 48 if rng.random() < 0.7 else 60
    L = int(rng.choice([100, 150, 151, 250]))
    G = int(rng.choice([500, 3000, 3000, 20000, 20000, 20000, 120000, 120000, 120000, 120000, 120000, 1000000]))
    cov = float(rng.choice([3, 8, 30, 60]))
    if G >= 1000000: cov = min(cov, 30.0)
    cap_pct = int(rng.choice([100, 100, 100, 60, 10]))       # shrink the partition's bucket capacity: overflow segment
    eng.set_option("msp_cap_pct", cap_pct)
    os.environ["SNK_TUNING"] = f"msp_cap_pct={cap_pct}"       # (the in-process ranks below create their own contexts)
    n = max(10, int(G * cov / L))
    err = float(rng.choice([0.0, 0.002, 0.01]))
    nbc = int(rng.choice([1, 3, 40]))
    min_freq = int(rng.choice([1, 2, 3, 4])); min_bc = int(rng.choice([0, 1, 2, 2, 3, 4]))
    nb = int(rng.choice([0, 0, 1, 5, 97, 4099]))
    use_bc = rng.random() < 0.8
    if os.environ.get("FUZZ_PROFILE") == "deep":      # few huge buckets, nothing filtered: deep hash splits, big sparse chunks
        G = int(rng.choice([120000, 300000])); cov = float(rng.choice([3, 8])); n = max(10, int(G * cov / L))
        min_freq = 1; nb = int(rng.choice([1, 2, 3, 5, 7])); err = float(rng.choice([0.0, 0.01]))
    codes, quals, lens, bc = make_reads(rng, G, n, L, err, nbc, rng.random() < 0.6)
    if only >= 0 and case != only:          # consume the same random draws as a full run of this case
        if K == 48 and rng.random() < 0.4:
            ng_ = int(rng.choice([1, 2, 7])); rng.integers(0, ng_, n); rng.choice([1, 1000])
        rng.choice([2, 3, 5, 8])
        continue
    if only >= 0:
        print(f"replaying case {case}: K={K} L={L} G={G} n={n} err={err} nbc={nbc} min_freq={min_freq} min_bc={min_bc} nb={nb} bc={use_bc} cap%={cap_pct}", flush=True)
    gl = oracle_lib.good_lens(quals, lens, K=K)
    if min_freq == 1 and use_bc and min_bc > 0:
        min_bc = 0          # without the prune (min_freq 1) a barcode filter leaves contexts that point at dropped k-mers:
                            # the reference's EdgeBuilder aborts on those ("failed to find k-mer"), so does the oracle
    try:
        o = oracle_lib.OracleResult(codes, gl, bc if use_bc else None, K=K, min_freq=min_freq, min_bc=min_bc, hbv=False)
    except RuntimeError as ex:
        print(f"skip case {case}: the oracle rejects it ({ex})", flush=True)
        continue
    rows = torch.from_numpy(synth.pack_rows(codes).view(np.int32)).to(dev)
    # two cases in three: quality rows padded to 4 bytes with garbage behind the read (the layout the readers produce): the trim
    # then runs inside the partition kernel; the others keep the bare rows (the separate trim kernel)
    quals_dev = quals
    if case % 3 != 0:
        prng = np.random.default_rng(1000 + case)
        quals_dev = prng.integers(0, 42, (quals.shape[0], (L + 3) // 4 * 4 + 4 * int(prng.integers(0, 2))), dtype=np.uint8)
        quals_dev[:, :L] = quals
    dq = torch.from_numpy(np.ascontiguousarray(quals_dev)).to(dev); dbc = torch.from_numpy(bc).to(dev) if use_bc else None
    dl = torch.from_numpy(lens.view(np.int16)).to(dev)
    tag = f"case {case}: K={K} L={L} G={G} n={n} err={err} nbc={nbc} min_freq={min_freq} min_bc={min_bc} nb={nb} bc={use_bc} cap%={cap_pct} -> {o.keys.shape[0]} k-mers, {len(o.unitigs)} unitigs"
    ok = True
    for glob in (0, 1):
        eng.set_option("global_graph", glob)
        if only >= 0: print("  single leg, global =", glob, flush=True)
        r = eng.count_graph(rows, L, quals=dq, bc=dbc, lens=dl, params=Params(K=K, min_freq=min_freq, min_bc=min_bc, n_buckets=nb))
        if not same(r.keys(), r.counts(), r.ctx(), r.unitigs(), o):
            ok = False; print("MISMATCH single", "global" if glob else "local", tag, flush=True)
        elif r.n_unitigs:
            # a14 on the device (snk_dev_hbv) against the host-array entry point on the BVComp-sorted unitigs
            from supernova_amd import graphio
            h = r.hbv()
            off_d, bases_d = r.unitig_arrays()
            lut = np.frombuffer(b"ACGT", dtype=np.uint8)
            ranked = [lut[bases_d[int(off_d[i]):int(off_d[i + 1])]].tobytes().decode() for i in h["order"]]
            off2, bases2 = graphio.unitigs_to_arrays(o.unitigs)
            h2 = graphio.hbv_from_unitigs(K, off2, bases2)
            if ranked != o.unitigs or any(not np.array_equal(h[kx], h2[kx]) for kx in ("v_left", "v_right", "src", "is_rc", "fwd", "rev")):
                ok = False; print("MISMATCH hbv", "global" if glob else "local", tag, flush=True)
    eng.set_option("global_graph", 0)
    if K == 48 and rng.random() < 0.4:
        # per-group graphs (BASELINE config 5): one grouped run == the oracle applied to every group's reads on its own
        NG = int(rng.choice([1, 2, 7]))
        group = rng.integers(0, NG, n).astype(np.int32) * int(rng.choice([1, 1000]))
        if only >= 0: print("  grouped leg, NG =", NG, flush=True)
        r = eng.count_graph(rows, L, quals=dq, bc=None, lens=dl, group=torch.from_numpy(group).to(dev),
                            params=Params(K=48, min_freq=min_freq, min_bc=0, grouped=True, sorted_table=False, n_buckets=nb))
        k, cnt, ctx = r.keys(), r.counts(), r.ctx()
        off, bases = r.unitig_arrays(); ug = r.unitig_groups()
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        gok = bool(np.all(np.diff(ug.astype(np.int64)) >= 0))
        tot = 0
        for gid in np.unique(group):
            sel = group == gid
            og = oracle_lib.OracleResult(codes[sel], gl[sel], None, K=48, min_freq=min_freq, min_bc=0, hbv=False)
            m = k[:, 3] == gid
            kk, cc, xx = k[m], cnt[m], ctx[m]
            order = np.lexsort((kk[:, 2], kk[:, 1], kk[:, 0]))
            us = sorted((lut[bases[int(off[u]):int(off[u + 1])]].tobytes().decode() for u in np.nonzero(ug == gid)[0]), key=lambda t: (-len(t), t))
            gok &= (kk.shape[0] == og.keys.shape[0] and np.array_equal(kk[order][:, :3], og.keys[:, :3]) and np.array_equal(cc[order], og.counts)
                    and np.array_equal(xx[order], og.ctx) and us == og.unitigs)
            tot += int(m.sum())
        gok &= tot == k.shape[0]
        if not gok:
            ok = False; print("MISMATCH grouped NG=%d" % NG, tag, flush=True)
    W = int(rng.choice([2, 3, 5, 8]))
    if only >= 0: print("  sharded leg, W =", W, flush=True)
    world = SimWorld(W); bounds = [n * q // W for q in range(W + 1)]; out = [None] * W; errs = []
    def worker(q):
        try:
            torch.cuda.set_device(0)
            e = Engine(0); lo, hi = bounds[q], bounds[q + 1]
            sh = ShardedEngine(e, world.comm(q))
            rr = sh.count_graph(rows[lo:hi].contiguous(), L, quals=dq[lo:hi].contiguous(), bc=None if dbc is None else dbc[lo:hi].contiguous(),
                                lens=dl[lo:hi].contiguous(), params=Params(K=K, min_freq=min_freq, min_bc=min_bc, n_buckets=(nb // W + 1) * W if nb else 0),
                                read_index_base=lo)
            out[q] = (rr.keys(), rr.counts(), rr.ctx(), rr.unitigs()); e.close()
        except BaseException as ex:
            errs.append(ex); world.barrier_obj.abort()
    ts = [threading.Thread(target=worker, args=(q,)) for q in range(W)]
    [t.start() for t in ts]; [t.join() for t in ts]
    if errs:
        ok = False; print("ERROR sharded W=%d" % W, tag, repr(errs[0])[:200], flush=True)
    else:
        keys = np.concatenate([x[0] for x in out]); cnt = np.concatenate([x[1] for x in out]); ctx = np.concatenate([x[2] for x in out])
        order = np.lexsort((keys[:, 3], keys[:, 2], keys[:, 1], keys[:, 0]))
        out[0] = out[0][:3] + (sorted((u for x in out for u in x[3]), key=lambda s: (-len(s), s)),)      # every rank wrote the unitigs it owns
        if not same(keys[order], cnt[order], ctx[order], out[0][3], o):
            ok = False; print("MISMATCH sharded W=%d" % W, tag, flush=True)
            k2, c2, x2 = keys[order], cnt[order], ctx[order]
            print("   keys", k2.shape, o.keys.shape, "equal" if np.array_equal(k2, o.keys) else "DIFF",
                  "| counts", "equal" if k2.shape == o.keys.shape and np.array_equal(c2, o.counts) else "DIFF",
                  "| ctx", "equal" if k2.shape == o.keys.shape and np.array_equal(x2, o.ctx) else "DIFF",
                  "| unitigs", len(out[0][3]), len(o.unitigs), "equal" if out[0][3] == o.unitigs else "DIFF", flush=True)
            if out[0][3] != o.unitigs:
                a, b = set(out[0][3]), set(o.unitigs)
                print("   only in HIP:", len(a - b), "only in oracle:", len(b - a), "lens", sorted(len(x) for x in a - b)[:10], sorted(len(x) for x in b - a)[:10], flush=True)
    bad += 0 if ok else 1
    print(("ok   " if ok else "FAIL ") + tag + f" (sharded W={W})", flush=True)
print(f"{n_cases - bad} of {n_cases} cases bit-exact")
sys.exit(1 if bad else 0)
