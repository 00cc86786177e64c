#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference to have been compiled by
oracle/ref/build_ref.sh into oracle/_ref/snref_driver).  Every case = inputs + the reference's
outputs (good lengths, retained k-mer table with counts and pruned contexts, k-mer spectrum,
canonical unitigs, HBV built from them).  Inputs are data, not reference code: seeded synthetic
reads from libsnk's generator, or the hand-built adversarial read set below (numpy, fixed seed).

usage: python tests/golden/make_golden.py [case ...]
"""
from __future__ import annotations

import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import refio  # noqa: E402
from supernova_amd import synth  # noqa: E402

COMP = str.maketrans("ACGT", "TGCA")


def rc(s: str) -> str:
    return s.translate(COMP)[::-1]


def synth_case(n_reads, seed, error_free):
    sp = synth.synth_params(n_reads, seed=seed, error_free=error_free)
    rows, quals, bc = synth.synth_host(sp)
    lens = np.full(n_reads, sp.read_len, dtype=np.uint16)
    asc = synth.codes_to_ascii(synth.unpack_rows(rows, sp.read_len))
    return dict(lens=lens, ascii=asc, quals=quals, bc=bc, ign_bc_below=0,
                meta=dict(kind="synth", n_reads=n_reads, seed=seed, error_free=bool(error_free)))


def adversarial_case(seed=20260928, K=48):
    """Hand-built graph torture set: repeats, SNP bubbles, a circular plasmid, a planted palindromic
    48-mer, a short-period tandem repeat, a poly-A run, short/trimmed reads, N bases, bc 0 / -1."""
    rng = np.random.default_rng(seed)
    rnd = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    rep = rnd(500)
    x24 = rnd(24)
    pal = x24 + rc(x24)                      # palindromic 48-mer (== its own reverse complement)
    assert pal == rc(pal)
    tandem = "ACGTTGA" * 14                  # period 7, 98 bp
    polya = "A" * 80
    hapA = rnd(6000) + rep + rnd(4000) + pal + rnd(3000) + rep + rnd(2500) + tandem + rnd(3000) + polya + rnd(2500) + rep + rnd(4000)
    hb = list(hapA)
    for p in range(700, len(hb) - 700, 997):  # SNP bubbles
        hb[p] = "ACGT"[("ACGT".index(hb[p]) + 1 + int(rng.integers(0, 3))) % 4]
    hapB = "".join(hb)
    plasmid = rnd(700)
    plasmid2 = rnd(61)                        # tiny circle: 61 k-mers
    contigs = [(hapA, False, 22.0), (hapB, False, 22.0), (plasmid, True, 60.0), (plasmid2, True, 80.0), (rnd(900), False, 3.0)]
    reads, quals, bcs = [], [], []
    L = 150
    n_bc = 60
    for seqs, circular, cov in contigs:
        n = int(cov * len(seqs) / L)
        for _ in range(n):
            ln = L if rng.random() < 0.85 else int(rng.integers(30, L))
            if circular:
                s = int(rng.integers(0, len(seqs)))
                frag = (seqs * (2 + ln // len(seqs)))[s:s + ln]
            else:
                if len(seqs) < ln:
                    continue
                s = int(rng.integers(0, len(seqs) - ln + 1))
                frag = seqs[s:s + ln]
            if rng.random() < 0.5:
                frag = rc(frag)
            fb = list(frag)
            q = np.full(ln, 30, dtype=np.uint8)
            for i in range(ln):                      # sequencing errors with a low quality
                if rng.random() < 0.003:
                    fb[i] = "ACGT"[("ACGT".index(fb[i]) + 1 + int(rng.integers(0, 3))) % 4]
                    q[i] = 12
            r = rng.random()
            if r < 0.08:                             # Q2 tail
                t = int(rng.integers(1, 60))
                q[max(0, ln - t):] = 2
            elif r < 0.12:                           # a bad base somewhere in the middle
                q[int(rng.integers(0, ln))] = 3
            elif r < 0.14:                           # exactly K good bases at the front (B skips these: len<K+1)
                q[:] = 2
                q[:min(K, ln)] = 30
            if rng.random() < 0.02:
                fb[int(rng.integers(0, ln))] = "N"
            reads.append("".join(fb))
            quals.append(q)
            b = int(rng.integers(1, n_bc + 1))
            if rng.random() < 0.05:
                b = 0
            bcs.append(b)
    # a region seen by a single barcode only (dropped by the >=2 barcode rule) ...
    solo = rnd(400)
    for _ in range(40):
        s = int(rng.integers(0, len(solo) - L + 1))
        reads.append(solo[s:s + L]); quals.append(np.full(L, 30, np.uint8)); bcs.append(7)
    # ... and one seen only by unbarcoded reads
    solo0 = rnd(400)
    for _ in range(40):
        s = int(rng.integers(0, len(solo0) - L + 1))
        reads.append(solo0[s:s + L]); quals.append(np.full(L, 30, np.uint8)); bcs.append(0)
    order = rng.permutation(len(reads))
    # the first reads are "non-10x" (bc = -1 through ign_bc_below): a one-barcode region kept on frequency alone
    ign = rnd(400)
    pre_r, pre_q, pre_b = [], [], []
    for _ in range(30):
        s = int(rng.integers(0, len(ign) - L + 1))
        pre_r.append(ign[s:s + L]); pre_q.append(np.full(L, 30, np.uint8)); pre_b.append(9)
    reads = pre_r + [reads[i] for i in order]
    quals = pre_q + [quals[i] for i in order]
    bcs = pre_b + [bcs[i] for i in order]
    n = len(reads)
    asc = np.full((n, L), ord("A"), dtype=np.uint8)
    qa = np.zeros((n, L), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    for i, (r, q) in enumerate(zip(reads, quals)):
        lens[i] = len(r)
        asc[i, :len(r)] = np.frombuffer(r.encode(), dtype=np.uint8)
        qa[i, :len(r)] = q
    return dict(lens=lens, ascii=asc, quals=qa, bc=np.asarray(bcs, dtype=np.int32), ign_bc_below=len(pre_r),
                meta=dict(kind="adversarial", seed=seed))


def dups_case(n_reads=4000, seed=0x5EED00D0):
    """Synthetic pairs plus planted duplicate pairs (MarkDups): copies with the same qualities (ties ->
    artifactual duplicates), with lowered / raised qualities (the best copy wins), in the same / another / no barcode
    (inter-barcode rate), groups of two to four, one copy with a changed mate head (not a duplicate)."""
    c = synth_case(n_reads, seed, False)
    rng = np.random.default_rng(seed)
    asc, qa, bc, lens = [c["ascii"]], [c["quals"]], [c["bc"]], [c["lens"]]
    npairs = n_reads // 2
    for p in rng.choice(npairs, 160, replace=False):
        for _ in range(int(rng.integers(1, 4))):
            a = c["ascii"][2 * p:2 * p + 2].copy()
            q = c["quals"][2 * p:2 * p + 2].copy()
            b = c["bc"][2 * p:2 * p + 2].copy()
            kind = int(rng.integers(0, 6))
            if kind == 1: q[:, 100:] = np.maximum(q[:, 100:], 3) - 1          # a worse copy
            elif kind == 2: q[0, 50:60] = 40                                   # a better copy
            elif kind == 3: b[:] = int(rng.integers(1, 1 << 20))               # another barcode
            elif kind == 4: b[:] = 0                                           # no barcode
            elif kind == 5: a[1, 2] = ord("ACGT"[("ACGT".index(chr(a[1, 2])) + 1) % 4])   # another mate head for read 0
            asc.append(a); qa.append(q); bc.append(b); lens.append(c["lens"][2 * p:2 * p + 2])
    out = dict(c)
    out.update(ascii=np.concatenate(asc), quals=np.concatenate(qa), bc=np.concatenate(bc).astype(np.int32), lens=np.concatenate(lens),
               meta=dict(kind="synth+dups", n_reads=n_reads, seed=seed))
    return out


CASES = {
    "synth_4k_dups": lambda: dups_case(),
    "synth_2k_err": lambda: synth_case(2000, 0x5EED0001, False),
    "synth_6k_clean": lambda: synth_case(6000, 0x5EED0002, True),
    "synth_20k_err": lambda: synth_case(20000, 0x5EED0003, False),
    "adversarial": lambda: adversarial_case(),
}


K60_CASES = ("adversarial", "synth_20k_err")


def make(name: str) -> None:
    case = CASES[name]()
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        refio.write_snkrd(td / "in.snkrd", case["lens"], case["ascii"], case["quals"], case["bc"], case["ign_bc_below"])
        log = refio.run_ref(td / "in.snkrd", td / "out")
        d = refio.read_ref_dump(td / "out")
        if name in K60_CASES:
            # the reference's K=60 variant (BuildReadQGraph60.cc; frequency rule only, no barcode rule)
            log60 = refio.run_ref(td / "in.snkrd", td / "out60", K=60)
            d60 = refio.read_ref_dump(td / "out60", K=60)
            s60 = [l for l in log60.splitlines() if l.startswith("SNREF_DUMP")][-1]
            np.savez_compressed(
                GOLD / f"{name}_k60.npz", exp_goodlens=d60["goodlens"], exp_keys=d60["kmers"]["k"],
                exp_counts=d60["kmers"]["count"], exp_ctx=d60["kmers"]["ctx"],
                exp_unitigs=np.frombuffer("\n".join(d60["unitigs"]).encode(), dtype=np.uint8),
                exp_hbv=np.frombuffer(d60["hbv"].encode(), dtype=np.uint8),
                exp_ahbv=d60["a.hbv"], exp_ainv=d60["a.inv"],
                ref_summary=np.frombuffer(s60.encode(), dtype=np.uint8))
            print(f"{name}_k60: {s60}")
    summary = [l for l in log.splitlines() if l.startswith("SNREF_DUMP")][-1]
    codes = synth.ascii_to_codes(case["ascii"])
    out = GOLD / f"{name}.npz"
    np.savez_compressed(
        out,
        lens=case["lens"], rows=synth.pack_rows(codes), ascii_has_n=np.argwhere(case["ascii"] == ord("N")).astype(np.int32),
        quals=case["quals"], bc=case["bc"], ign_bc_below=np.int64(case["ign_bc_below"]),
        exp_goodlens=d["goodlens"], exp_keys=d["kmers"]["k"], exp_counts=d["kmers"]["count"], exp_ctx=d["kmers"]["ctx"],
        exp_unitigs=np.frombuffer("\n".join(d["unitigs"]).encode(), dtype=np.uint8),
        exp_hbv=np.frombuffer(d["hbv"].encode(), dtype=np.uint8),
        exp_hist=np.asarray(d["hist"]["vals"] if d["hist"] else [], dtype=np.int64),
        exp_path_off=d["path_off"], exp_path_n=d["path_n"], exp_path_edges=d["path_edges"],
        exp_ahbv=d["a.hbv"], exp_ainv=d["a.inv"],
        exp_dup=d["dup"] if d["dup"] is not None else np.zeros(0, np.uint8),
        exp_interdup=np.float64(d.get("interdup", 0.0)), exp_art_perc=np.float64(d.get("art_perc", 0.0)),
        meta=np.frombuffer(repr(case["meta"]).encode(), dtype=np.uint8),
        ref_summary=np.frombuffer(summary.encode(), dtype=np.uint8),
    )
    print(f"{name}: {summary}  -> {out.name} ({out.stat().st_size/1024:.0f} KiB)")


GOLD = Path(__file__).resolve().parent


def make_formats() -> None:
    """tests/golden/formats/: the ASSEMBLER_DF stage inputs (reads.fastb / reads.qualp / reads.bci) of the
    synth_2k_err reads sorted by barcode, written by the REFERENCE's own writers (vecbvec::WriteAll,
    ObjectManager<VecPQVec>::store, BinaryWriter::writeFile) through `snref_driver ... formats`."""
    import shutil
    case = CASES["synth_2k_err"]()
    order = np.argsort(case["bc"], kind="stable")
    out = GOLD / "formats"
    out.mkdir(exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        refio.write_snkrd(td / "in.snkrd", case["lens"][order], case["ascii"][order], case["quals"][order], case["bc"][order])
        print(refio.run_ref(td / "in.snkrd", td / "out", mode="formats").splitlines()[-1])
        for f in ("reads.fastb", "reads.qualp", "reads.bci"):
            shutil.copy(td / "out" / f, out / f)
    np.save(out / "order.npy", order.astype(np.int32))


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES) + ["formats"]
    for nm in names:
        make_formats() if nm == "formats" else make(nm)
