"""Pin the CPU oracle (oracle/snk_oracle.c) against the reference's own outputs and known-answer tests.

Golden vectors: tests/golden/*.npz, dumped from the reference binary (path B, lib/assembly) by
tests/golden/make_golden.py.  Known-answer tests restated from the reference's Rust unit tests.
"""
import numpy as np
import pytest

import goldens
import oracle_lib


@pytest.mark.parametrize("name", goldens.CASES)
def test_oracle_matches_reference(name):
    c = goldens.load(name)
    gl = oracle_lib.good_lens(c.quals, c.lens, K=48, min_qual=7)
    assert np.array_equal(gl, c.exp_goodlens)
    o = oracle_lib.OracleResult(c.codes, gl, c.bc, K=48, min_freq=3, min_bc=2, ign_bc_below=c.ign_bc_below)
    assert o.keys.shape[0] == c.exp_keys.shape[0]
    assert np.array_equal(o.keys[:, :3], c.exp_keys)
    assert np.all(o.keys[:, 3] == 0)
    assert np.array_equal(np.minimum(o.counts, (1 << 24) - 1), c.exp_counts)   # B saturates at 2^24-1 (ReadPather.h:128-129)
    assert np.array_equal(o.ctx, c.exp_ctx)
    assert o.unitigs == c.exp_unitigs
    assert o.hbv_text() == c.exp_hbv
    hist = np.bincount(np.minimum(o.counts, (1 << 24) - 1)) if len(o.counts) else np.zeros(0, np.int64)
    assert np.array_equal(hist, c.exp_hist)       # stats/histogram_kmer_count.json (BuildReadQGraph48.cc:199-216)


def test_qv_trim_read_kat():
    """lib/tada/src/cmd_msp.rs:329-350 test_qv_trim_read (min_qual 10, K=48), expectation formula at :347."""
    quals = np.frombuffer(
        b"FFFFFFFFIFFFFFFFFFFIIIFFBFIFFFFIFBFFIBFIFBBFFIFFIFFFFFFFFFFFBBBBBBBBBB07BB7BB<BBBBBBBBBB", dtype=np.uint8)
    K = 48
    n = len(quals)
    for i in range(n):
        q = quals.copy()
        q[i] = 34
        got = int(oracle_lib.good_lens((q - 33)[None, :], n, K=K, min_qual=10)[0])
        exp = n if i < n - K else (i if i >= K else 0)
        assert got == exp, (i, got, exp)


def test_msp_slices_cover_all_kmers():
    """lib/tada/src/msp/mod.rs:202-220 test_slice: union of k-mers over MSP slices == all k-mers (k=50,p=8)."""
    rng = np.random.default_rng(7)
    k, p = 50, 8
    perm = rng.permutation(1 << (2 * p)).astype(np.uint32)
    for it in range(50):
        seq = rng.integers(0, 4, int(rng.integers(k, 400)), dtype=np.uint8)
        sl = oracle_lib.msp_scan(k, p, seq, perm)
        s = bytes(seq)
        allk = {s[i:i + k] for i in range(len(s) - k + 1)}
        got = set()
        for (_v, mp, st, ln) in sl:
            assert ln >= k and ln <= 2 * k - p
            assert st <= mp and mp + p <= st + ln
            for i in range(st, st + ln - k + 1):
                got.add(s[i:i + k])
        assert got == allk


def test_msp_shard_consistency():
    """lib/tada/src/kmer/mod.rs:1102-1150 check_consistent_shard: a k-mer (either strand) always gets the same value."""
    rng = np.random.default_rng(11)
    k, p = 48, 8
    perm = rng.permutation(1 << (2 * p)).astype(np.uint32)
    seen = {}
    base = rng.integers(0, 4, 300, dtype=np.uint8)
    for it in range(60):
        seq = base.copy()
        for _ in range(3):
            seq[int(rng.integers(0, len(seq)))] = rng.integers(0, 4)
        if it & 1:
            seq = (3 - seq[::-1]).astype(np.uint8)
        s = bytes(seq)
        for (v, _mp, st, ln) in oracle_lib.msp_scan(k, p, seq, perm):
            for i in range(st, st + ln - k + 1):
                km = s[i:i + k]
                rc = bytes(3 - b for b in reversed(km))
                key = min(km, rc)
                assert seen.setdefault(key, v) == v


def test_bv_roundtrip(tmp_path):
    """.bv hand-off format round trip (lib/tada/src/sim_tests.rs:142-179; debruijn.rs:895-929)."""
    import ctypes as C
    lib = oracle_lib.load()
    c = goldens.load("adversarial")
    o = oracle_lib.OracleResult(c.codes, c.exp_goodlens, c.bc, ign_bc_below=c.ign_bc_below, hbv=False)
    u = oracle_lib.Unitigs()
    off = np.ascontiguousarray(o.unitig_off, dtype=np.uint64)
    b = np.ascontiguousarray(o.unitig_bases, dtype=np.uint8)
    u.n = len(o.unitigs)
    u.off = off.ctypes.data_as(C.POINTER(C.c_uint64))
    u.bases = b.ctypes.data_as(C.POINTER(C.c_uint8))
    path = str(tmp_path / "x.bv").encode()
    assert lib.sno_write_bv(path, C.byref(u)) == 0
    raw = open(path, "rb").read()
    assert raw[:8] == b"BINWRITE" and int.from_bytes(raw[8:16], "little") == len(o.unitigs)
    v = oracle_lib.Unitigs()
    assert lib.sno_read_bv(path, C.byref(v)) == 0
    assert v.n == u.n
    off2 = np.ctypeslib.as_array(v.off, shape=(v.n + 1,))
    assert np.array_equal(off2, off)
    assert np.array_equal(np.ctypeslib.as_array(v.bases, shape=(int(off[-1]),)), b)
    lib.sno_unitigs_free(C.byref(v))


@pytest.mark.parametrize("name", goldens.K60_CASES)
def test_oracle_k60_matches_reference(name):
    """K=60 pin: the reference's BuildReadQGraph60 (frequency rule only -> oracle run without a barcode vector)."""
    g = goldens.Case60(name)
    c = g.base
    gl = oracle_lib.good_lens(c.quals, c.lens, K=60, min_qual=7)
    assert np.array_equal(gl, g.exp_goodlens)
    o = oracle_lib.OracleResult(c.codes, gl, None, K=60, min_freq=3)
    assert np.array_equal(o.keys, g.exp_keys)
    assert np.array_equal(np.minimum(o.counts, (1 << 24) - 1), g.exp_counts)
    assert np.array_equal(o.ctx, g.exp_ctx)
    assert o.unitigs == g.exp_unitigs
    assert o.hbv_text() == g.exp_hbv


@pytest.mark.parametrize("name", goldens.CASES)
def test_read_paths_match_reference(name):
    """f1: the restatement of pathReads (new aligner) against the read paths the reference binary dumped for the golden cases:
    offset and HBV edge ids of every read, untrimmed reads, N bases as A."""
    c = goldens.load(name)
    off, n, edges = oracle_lib.path_reads(c.codes, c.quals, c.lens, c.exp_unitigs, K=48)
    bad = np.nonzero((n != c.exp_path_n))[0]
    assert len(bad) == 0, (len(bad), bad[:5], n[bad[:5]], c.exp_path_n[bad[:5]])
    assert np.array_equal(edges, c.exp_path_edges)
    assert np.array_equal(off, c.exp_path_off)


def _art_matches_log(n_art, n_pairs, logged):
    """The reference only LOGS the artifactual-duplicate percentage (two significant digits after its fixed/precision
    manipulators): the count is checked through the same rounding."""
    return float(f"{100.0 * n_art / n_pairs:.2g}") == float(f"{logged:.2g}")


@pytest.mark.parametrize("name", goldens.CASES)
def test_mark_dups_matches_reference(name):
    """f4: the restatement of MarkDups against the reference's own MarkDups (10X/SecretOps.cc:413-593) run on the read paths
    it made for the golden cases: the flag of every pair, the inter-barcode rate exactly, the artifact share as logged."""
    c = goldens.load(name)
    dup, art, rate, nd, ni = oracle_lib.mark_dups(c.codes, c.quals, c.lens, c.exp_path_off, c.exp_path_n, c.exp_path_edges, bc=c.bc)
    assert len(dup) == len(c.exp_dup)
    assert np.array_equal(dup, c.exp_dup), np.nonzero(dup != c.exp_dup)[0][:10]
    assert rate == c.exp_interdup
    assert _art_matches_log(int(art.sum()), len(dup), c.exp_art_perc), (int(art.sum()), len(dup), c.exp_art_perc)
    if name == "synth_4k_dups":
        assert dup.sum() > 300 and art.sum() > 100 and 0.0 < rate < 1.0


def test_mark_dups_hand_case():
    """MarkDups by hand (10X/SecretOps.cc:413-593): four pairs of 10-base reads.  Reads 0 and 2 sit on (edge 5, offset 3) and
    their mates start with the same five bases: one duplicate group; read 2's pair has the larger quality sum and survives, so
    pair 0 is flagged; their barcodes differ (inter-barcode rate 1).  Read 4 shares the placement but its mate starts
    differently: no duplicate.  Reads 6 and 7... are not placed.  Then the same with equal qualities: a tie, the earliest read
    survives (pair 1 flagged) and pair 1 is an artifactual duplicate (identical bases and qualities)."""
    n, L = 8, 10
    codes = np.zeros((n, L), dtype=np.uint8)
    codes[1] = [0, 1, 2, 3, 0, 1, 1, 1, 1, 1]
    codes[3] = [0, 1, 2, 3, 0, 2, 2, 2, 2, 2]            # same first five bases as read 1
    codes[5] = [3, 1, 2, 3, 0, 1, 1, 1, 1, 1]            # another head
    quals = np.full((n, L), 5, dtype=np.uint8)
    quals[2] = 7                                          # pair 1: sum 120 against pair 0's 100
    path_off = np.array([3, 0, 3, 0, 3, 0, 0, 0], dtype=np.int32)
    path_n = np.array([1, 0, 1, 0, 1, 0, 0, 0], dtype=np.int32)
    path_edges = np.array([5, 5, 5], dtype=np.int32)
    bc = np.array([7, 7, 9, 9, 7, 7, 0, 0], dtype=np.int32)
    dup, art, rate, nd, ni = oracle_lib.mark_dups(codes, quals, L, path_off, path_n, path_edges, bc=bc)
    assert dup.tolist() == [1, 0, 0, 0] and art.tolist() == [0, 0, 0, 0] and (rate, nd, ni) == (1.0, 1, 1)
    quals[2] = 5                                          # a tie: the earliest read wins, the later identical one is an artifact
    bc[2:4] = 7
    dup, art, rate, nd, ni = oracle_lib.mark_dups(codes, quals, L, path_off, path_n, path_edges, bc=bc)
    assert dup.tolist() == [0, 1, 0, 0] and art.tolist() == [0, 1, 0, 0] and (rate, nd, ni) == (0.0, 1, 0)
    # barcode 0 adopts the next member's barcode without a comparison (:463-466)
    bc[0:2] = 0
    bc[2:4] = 9
    _, _, rate, _, _ = oracle_lib.mark_dups(codes, quals, L, path_off, path_n, path_edges, bc=bc)
    assert rate == 0.0
    # no barcode vector at all, and nothing placed
    dup, art, rate, nd, ni = oracle_lib.mark_dups(codes, quals, L, path_off, np.zeros(n, np.int32), np.zeros(0, np.int32))
    assert dup.sum() == 0 and art.sum() == 0 and (rate, nd, ni) == (0.0, 0, 0)


@pytest.mark.parametrize("name", ["robust_repeats_200k", "robust_err15_200k"])
def test_oracle_matches_reference_digest_off_the_operating_point(name):
    """The C restatement against the REFERENCE's digests (tests/golden/big_hashes.json, made by make_big_hashes.py from
    oracle/_ref/snref_driver) on 200 k reads of the models bench.py's config.robust runs: a repeat-rich genome (interspersed families at
    1-3 % divergence, exact 5-kb duplications, STRs, poly-A) and 1.5 % sequencing errors."""
    import json
    from pathlib import Path
    import bighash
    from supernova_amd import synth
    fix = Path(__file__).resolve().parent / "golden" / "big_hashes.json"
    exp = json.loads(fix.read_text()).get(name)
    if exp is None:
        pytest.skip("no fixture")
    sp = synth.synth_params(exp["n_reads"], seed=exp["seed"], **exp.get("overrides", {}))
    rows, quals, bc = synth.synth_host(sp)
    gl = oracle_lib.good_lens(quals, sp.read_len)
    o = oracle_lib.OracleResult(synth.unpack_rows(rows, sp.read_len), gl, bc, hbv=False)
    hist = np.bincount(np.minimum(o.counts, (1 << 24) - 1)).astype(np.int64)
    dg = bighash.digest(gl.astype(np.uint32), o.keys, np.minimum(o.counts, (1 << 24) - 1), o.ctx, o.unitigs, hist, kw=3)
    bad = [f for f in ("n_reads", "n_kmers", "n_unitigs", "unitig_bases", "goodlens", "keys", "counts", "ctx", "hist", "unitigs") if dg[f] != exp[f]]
    assert not bad, {f: (dg[f], exp[f]) for f in bad}
