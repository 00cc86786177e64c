"""GPU parity: the HIP path (through the C ABI) against the reference's golden vectors and the CPU oracle.

Bar: bit-exact (integer/byte work).  Golden cases were dumped from the reference binary; seeded synthetic
cases are checked against oracle/snk_oracle.c on the same inputs.
"""
import numpy as np
import pytest

import goldens
import oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["local", "global"], autouse=True)
def graph_stage(request, monkeypatch, tune):
    """Every parity test runs twice: bucket-local graph stage (default) and the global one (cross-check)."""
    tune("SNK_GLOBAL_GRAPH", "1" if request.param == "global" else "0")
    return request.param


@pytest.fixture(scope="module")
def engine(snk):
    import torch
    from supernova_amd.engine import Engine
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    e = Engine(0)
    yield e
    e.close()


def _to_dev(c):
    import torch
    dev = torch.device("cuda", 0)
    rows = torch.from_numpy(c.rows.view(np.int32)).to(dev)
    quals = torch.from_numpy(np.ascontiguousarray(c.quals)).to(dev)
    bc = torch.from_numpy(c.bc.astype(np.int32)).to(dev)
    lens = torch.from_numpy(c.lens.astype(np.uint16).view(np.int16)).to(dev)
    return rows, quals, bc, lens


def _check_against(res, keys3, counts, ctx, unitigs, goodlens, hist):
    assert np.array_equal(res.good_len().astype(np.uint32), goodlens)
    k = res.keys()
    assert k.shape[0] == keys3.shape[0], (k.shape, keys3.shape)
    assert np.array_equal(k[:, :3], keys3)
    assert np.all(k[:, 3] == 0)
    assert np.array_equal(np.minimum(res.counts(), (1 << 24) - 1), counts)
    assert np.array_equal(res.ctx(), ctx)
    spec = res.spectrum()
    nz = np.nonzero(spec)[0]
    h = spec[: (nz[-1] + 1 if len(nz) else 0)].astype(np.int64)
    assert np.array_equal(h, hist)
    assert res.unitigs() == unitigs


@pytest.mark.parametrize("long_minimiser", [False, True])
@pytest.mark.parametrize("name", goldens.CASES)
def test_golden_case(engine, name, long_minimiser):
    """long_minimiser: SNK_F_LONG_MINIMISER -- 20-base minimisers in a 64-bit rolling window (genomes of human size); which minimiser cuts the
    reads is internal: same table, contexts, spectrum and unitigs."""
    from supernova_amd.engine import Params
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48, long_minimiser=long_minimiser),
                             ign_bc_below=c.ign_bc_below)
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)


@pytest.mark.parametrize("name", goldens.CASES)
def test_golden_case_strided_fragment_copy(engine, name, tune):
    """The join's fragment copy with a grid of eight workgroups: the kernel strides over the fragments (as it has to from 2^29 fragments on,
    where eight lanes per fragment are more work items than a launch may have -- a launch that was cut short silently at 750 M reads on
    one GPU, tools/r6_full_job.py)."""
    from supernova_amd.engine import Params
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    tune("emit_grid_log2", 3)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48), ign_bc_below=c.ign_bc_below)
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)


# every count-kernel instantiation the library can pick BY ITSELF, forced: env -> the usable table slots the call must report
COUNT_VARIANTS = {"screen": ({"SNK_COUNT_SCREEN_NG": "2"}, 960),      # bit filter + 1024-slot table (error-rich data: snk_count.hip SCREEN)
                  "tight": ({"SNK_COUNT_TIGHT": "1920", "SNK_COUNT_SCREEN_NG": "0"}, 1920)}      # booked slots, no filter


@pytest.mark.parametrize("variant", sorted(COUNT_VARIANTS))
@pytest.mark.parametrize("name", goldens.CASES)
def test_golden_case_count_variants(engine, monkeypatch, name, variant, tune):
    """test_golden_case with the ungrouped SCREEN / TIGHT count kernels forced (the kernels error-rich production data
    take were compared in-suite with the default kernel only): the reference's goldens, and the call reports which kernel ran."""
    from supernova_amd.engine import Params
    env, limit = COUNT_VARIANTS[variant]
    for k, v in env.items():
        tune(k, v)
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48), ign_bc_below=c.ign_bc_below)
    assert engine.last_count_limit() == limit
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)


@pytest.mark.parametrize("variant", sorted(COUNT_VARIANTS))
@pytest.mark.parametrize("n_reads,error_free", [(200_000, False), (100_000, True)])
def test_synth_vs_oracle_count_variants(engine, monkeypatch, n_reads, error_free, variant, tune):
    from supernova_amd import synth
    from supernova_amd.engine import Params
    env, limit = COUNT_VARIANTS[variant]
    for k, v in env.items():
        tune(k, v)
    sp = synth.synth_params(n_reads, seed=0x5EED0100 + n_reads % 97, error_free=error_free)
    rows_h, quals_h, bc_h = synth.synth_host(sp, qstride=160)
    rows_d, quals_d, bc_d = engine.synth(sp, qstride=160)
    res = engine.count_graph(rows_d, 150, quals=quals_d, bc=bc_d, params=Params(K=48))
    assert engine.last_count_limit() == limit
    gl = oracle_lib.good_lens(quals_h, 150)
    o = oracle_lib.OracleResult(synth.unpack_rows(rows_h, 150), gl, bc_h, hbv=False)
    hist = np.bincount(np.minimum(o.counts, (1 << 24) - 1)).astype(np.int64)
    _check_against(res, o.keys[:, :3], np.minimum(o.counts, (1 << 24) - 1), o.ctx, o.unitigs, gl, hist)


@pytest.mark.parametrize("n_buckets", [1, 7, 4096])
def test_bucket_count_independence(engine, n_buckets):
    """Shard assignment is internal: any bucket count gives the same table, including the
    forced LDS-table overflow / split path (n_buckets=1)."""
    from supernova_amd.engine import Params
    c = goldens.load("adversarial")
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48, n_buckets=n_buckets),
                             ign_bc_below=c.ign_bc_below)
    if n_buckets == 1:
        assert res.buckets_split >= 1
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)


@pytest.mark.parametrize("case,K,n_buckets,slots", [("adversarial", 48, 1, 1920), ("adversarial", 48, 7, 300), ("synth_20k_err", 48, 3, 1920),
                                                    ("synth_20k_err", 48, 16, 700), ("synth_20k_err", 60, 2, 1920), ("synth_20k_err", 48, 0, 1984)])
def test_booked_table_slots_count_the_same_table(engine, monkeypatch, case, K, n_buckets, slots, tune):
    """The count kernel's second variant (snk_count.hip, TIGHT: waves book their slots, the table fills to `slots` of 2048 instead of
    1216, a pass that retains more than one graph chunk's worth is counted again in halves) gives the table and the unitigs of the
    default one -- with buckets that overflow and split (few buckets), with a small limit (bookings fail all the time), at K=60."""
    from supernova_amd.engine import Params
    g = goldens.Case60(case) if K == 60 else goldens.load(case)
    c = g.base if K == 60 else g
    rows, quals, bc, lens = _to_dev(c)
    tune("SNK_COUNT_TIGHT", str(slots))
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc if K == 48 else None, lens=lens, params=Params(K=K, n_buckets=n_buckets),
                             ign_bc_below=c.ign_bc_below)
    assert engine.last_count_limit() == slots
    if n_buckets == 1:
        assert res.buckets_split >= 1
    if K == 48:
        _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)
    else:
        assert np.array_equal(res.keys(), g.exp_keys) and np.array_equal(np.minimum(res.counts(), (1 << 24) - 1), g.exp_counts)
        assert np.array_equal(res.ctx(), g.exp_ctx) and res.unitigs() == g.exp_unitigs
    tune("SNK_COUNT_TIGHT", "0")
    engine.count_graph(rows, c.read_len, quals=quals, bc=bc if K == 48 else None, lens=lens, params=Params(K=K, n_buckets=n_buckets), ign_bc_below=c.ign_bc_below)
    assert engine.last_count_limit() == 1216


@pytest.mark.parametrize("n_reads,error_free", [(200_000, False), (100_000, True)])
def test_synth_vs_oracle(engine, n_reads, error_free):
    """Device generator == host generator, and the full path == oracle on a seeded workload of the bench's shape."""
    from supernova_amd import synth
    from supernova_amd.engine import Params
    sp = synth.synth_params(n_reads, seed=0x5EED0100 + n_reads % 97, error_free=error_free)
    rows_h, quals_h, bc_h = synth.synth_host(sp, qstride=160)
    rows_d, quals_d, bc_d = engine.synth(sp, qstride=160)
    assert np.array_equal(rows_d.cpu().numpy().view(np.uint32), rows_h)
    assert np.array_equal(quals_d.cpu().numpy()[:, :150], quals_h[:, :150])
    assert np.array_equal(bc_d.cpu().numpy(), bc_h)
    res = engine.count_graph(rows_d, 150, quals=quals_d, bc=bc_d, params=Params(K=48))
    gl = oracle_lib.good_lens(quals_h, 150)
    o = oracle_lib.OracleResult(synth.unpack_rows(rows_h, 150), gl, bc_h, hbv=False)
    assert res.n_instances == o.n_instances
    hist = np.bincount(np.minimum(o.counts, (1 << 24) - 1)).astype(np.int64)
    _check_against(res, o.keys[:, :3], np.minimum(o.counts, (1 << 24) - 1), o.ctx, o.unitigs, gl, hist)


@pytest.mark.parametrize("long_minimiser", [False, True])
def test_crowded_minimiser_space_vs_oracle(engine, long_minimiser):
    """repeat_mode bit 4: every fifth base of the genome is A -- ~1 site per canonical 16-mer value at the bench's size, the minimiser-sharing
    regime of a human genome.  Both minimiser lengths give the oracle's table and unitigs."""
    from supernova_amd import synth
    from supernova_amd.engine import Params
    sp = synth.synth_params(120_000, seed=77, repeat_mode=16)
    rows_h, quals_h, bc_h = synth.synth_host(sp, qstride=160)
    rows_d, quals_d, bc_d = engine.synth(sp, qstride=160)
    assert np.array_equal(rows_d.cpu().numpy().view(np.uint32), rows_h)          # the mode is the shared generator's: device == host
    codes = synth.unpack_rows(rows_h, 150)
    assert all(any((r[p::5] == 0).mean() > 0.9 for p in range(5)) or any((r[p::5] == 3).mean() > 0.9 for p in range(5)) for r in codes[:200])
    res = engine.count_graph(rows_d, 150, quals=quals_d, bc=bc_d, params=Params(K=48, long_minimiser=long_minimiser))
    gl = oracle_lib.good_lens(quals_h, 150)
    o = oracle_lib.OracleResult(codes, gl, bc_h, hbv=False)
    hist = np.bincount(np.minimum(o.counts, (1 << 24) - 1)).astype(np.int64)
    _check_against(res, o.keys[:, :3], np.minimum(o.counts, (1 << 24) - 1), o.ctx, o.unitigs, gl, hist)


def test_no_barcodes_and_minbc_modes(engine):
    """bc == NULL disables the barcode rule (BuildReadQGraph48.cc:176-178); min_bc 0/1 follow areEnoughBarcodes."""
    import torch
    from supernova_amd.engine import Params
    c = goldens.load("adversarial")
    rows, quals, bc, lens = _to_dev(c)
    gl = c.exp_goodlens
    # min_bc > 2: areEnoughBarcodes counts distinct barcodes for any minBC (BuildReadQGraph48.cc:117-137) -- per-slot id sets
    for min_bc, bcarg in [(2, None), (0, bc), (1, bc), (3, bc), (4, bc), (6, bc), (8, bc)]:
        res = engine.count_graph(rows, c.read_len, quals=quals, bc=bcarg, lens=lens, params=Params(K=48, min_bc=min_bc),
                                 ign_bc_below=c.ign_bc_below)
        o = oracle_lib.OracleResult(c.codes, gl, None if bcarg is None else c.bc, min_bc=min_bc,
                                    ign_bc_below=c.ign_bc_below, hbv=False)
        hist = np.bincount(np.minimum(o.counts, (1 << 24) - 1)).astype(np.int64)
        _check_against(res, o.keys[:, :3], o.counts, o.ctx, o.unitigs, gl, hist)


def test_pack_ascii_and_trim_kernels(engine):
    import torch
    from supernova_amd import synth
    c = goldens.load("adversarial")
    asc = synth.codes_to_ascii(c.codes).copy()
    asc[3, 5] = ord("N")
    asc[9, 0] = ord("n")
    dev = torch.device("cuda", 0)
    rows = engine.pack_ascii(torch.from_numpy(asc).to(dev), c.read_len).cpu().numpy().view(np.uint32)
    codes = c.codes.copy()
    codes[3, 5] = 0
    codes[9, 0] = 0
    assert np.array_equal(rows, synth.pack_rows(codes))
    quals = torch.from_numpy(np.ascontiguousarray(c.quals)).to(dev)
    lens = torch.from_numpy(c.lens.astype(np.uint16).view(np.int16)).to(dev)
    for mq in (7, 10, 31):
        g = engine.trim(quals, c.read_len, K=48, min_qual=mq, lens=lens).cpu().numpy().view(np.uint16)
        assert np.array_equal(g.astype(np.uint32), oracle_lib.good_lens(c.quals, c.lens, K=48, min_qual=mq))
    # rows padded to a multiple of 4 bytes take the LDS-tiled kernel: ragged lengths, a row count that is not a
    # multiple of the 256-row tile, garbage in the padding
    rng = np.random.default_rng(5)
    for n in (1, 255, 257, 1000, len(c.lens)):
        qp = rng.integers(0, 41, (n, 152), dtype=np.uint8)
        qp[:, :c.read_len] = c.quals[:n]
        for mq, K in ((7, 48), (10, 60)):
            g = engine.trim(torch.from_numpy(qp).to(dev), c.read_len, K=K, min_qual=mq, lens=lens[:n].contiguous()).cpu().numpy().view(np.uint16)
            assert np.array_equal(g.astype(np.uint32), oracle_lib.good_lens(c.quals[:n], c.lens[:n], K=K, min_qual=mq))


@pytest.mark.parametrize("K", [48, 60])
def test_trim_inside_the_partition_kernel(engine, K, monkeypatch, tune):
    """Quality rows padded to 4 bytes are trimmed by the partition kernel itself (snk_msp.hip, fused trim): the good lengths
    and the instance count it reports must be the trim kernel's / the oracle's (GoodLenTailFinder, BuildReadQGraph48.cc:65-89)
    on clean reads (decided by their last K quals), on reads with low-quality tails and on rows of noise (the bit-mask scan),
    with ragged lengths, and the counted table must not depend on where the trim ran."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Params
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(17 + K)
    n, L = 3001, 150
    codes = rng.integers(0, 4, (n, L), dtype=np.uint8)
    codes[1000:] = codes[rng.integers(0, 1000, n - 1000)]           # repeated reads: something survives min_freq
    rows = torch.from_numpy(synth.pack_rows(codes).view(np.int32)).to(dev)
    quals = np.full((n, 152), 30, dtype=np.uint8)
    quals[:, 150:] = rng.integers(0, 3, (n, 2))                      # padding: garbage below every threshold
    kind = rng.integers(0, 6, n)
    for i in range(n):
        if kind[i] == 1: quals[i, L - rng.integers(1, 120):L] = 2                       # a low-quality tail
        elif kind[i] == 2: quals[i, :L] = rng.integers(0, 41, L)                        # noise
        elif kind[i] == 3: quals[i, rng.integers(0, L, 3)] = rng.integers(0, 12, 3)     # a few dips
        elif kind[i] == 4: quals[i, L - K - rng.integers(0, 4)] = 0                     # a dip right at the window's edge
    lens = np.full(n, L, dtype=np.uint16)
    ragged = rng.random(n) < 0.3
    lens[ragged] = rng.integers(0, L + 1, int(ragged.sum()))
    lens[:4] = (0, K - 1, K, K + 1)
    qd = torch.from_numpy(quals).to(dev)
    ld = torch.from_numpy(lens.view(np.int16)).to(dev)
    for mq in (0, 7, 20, 31, 255):
        want = oracle_lib.good_lens(quals[:, :L], lens, K=K, min_qual=mq)
        params = Params(K=K, min_freq=2, min_qual=mq)
        got = {}
        for fused in ("1", "0"):
            tune("SNK_TRIM_FUSED", fused)
            res = engine.count_graph(rows, L, quals=qd, lens=ld, params=params)
            gl = res.good_len().astype(np.uint32)
            assert np.array_equal(gl, want), (mq, fused, np.flatnonzero(gl != want)[:8])
            assert res.n_instances == int(np.where(want >= K + 1, want - K + 1, 0).sum())
            keys, counts = res.keys(), res.counts()
            order = np.lexsort(tuple(keys[:, j] for j in range(keys.shape[1])))
            got[fused] = (keys[order], counts[order], res.n_unitigs)
        assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1], got["0"][1]) and got["1"][2] == got["0"][2]
        g2 = engine.trim(qd, L, K=K, min_qual=mq, lens=ld).cpu().numpy().view(np.uint16)
        assert np.array_equal(g2.astype(np.uint32), want)


def test_error_rich_reads_are_repartitioned_into_smaller_buckets(engine, monkeypatch, tune):
    """Bucket size follows the data: when the first buckets hold more distinct k-mers than the count kernel's LDS table takes
    (1.5 % errors here), the one-GPU path partitions a second time into smaller buckets instead of hash-splitting nearly every
    bucket, and later calls on the context start there.  The result does not depend on the bucket count: the same table and the
    same unitigs as with the fixed default size."""
    import math
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params
    n = 1_200_000
    sp = synth.synth_params(n, seed=0x5EED0E77, sub_ppm=15000, lowq_tail_ppm=200000)
    lam, term, cum = 150 * 15000 / 1e6, math.exp(-150 * 15000 / 1e6), 0.0
    for j in range(4):
        cum += term
        sp.err_cdf[j] = min(0xFFFFFFFF, int(cum * 4294967296.0))
        term *= lam / (j + 1)
    e = Engine(0)               # a fresh context: no bucket-size hint from earlier calls
    try:
        rows, quals, bc = e.synth(sp)

        def table(r):
            k, c, x = r.keys(), r.counts(), r.ctx()
            o = np.lexsort(tuple(k[:, j] for j in range(k.shape[1] - 1, -1, -1)))
            return k[o], c[o], x[o], sorted(r.unitigs())

        tune("SNK_ADAPTIVE_BUCKETS", "0")
        r0 = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
        assert r0.repartitioned == 0 and r0.buckets_split > r0.n_buckets // 2
        assert e.last_count_limit() == 1216          # (nothing known about the data: the default kernel)
        ref = table(r0)
        nb0 = r0.n_buckets
        tune("SNK_ADAPTIVE_BUCKETS", "1")
        tune("SNK_SCREEN_RATIO_PCT", "20")        # the filter's threshold: pinned, so that WHICH kernel runs is asserted, not either
        e2 = Engine(0)
        try:
            r1 = e2.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
            assert r1.repartitioned == 1 and r1.n_buckets > 1.1 * nb0 and r1.buckets_split < r1.n_buckets // 4
            # tables that run this full (0.4 distinct k-mers per instance): the second partition is counted behind the bit filter, whose table
            # has 960 usable slots (SNK_COUNT_SCREEN_NG=0: booked slots alone, 1920, and buckets half the size)
            assert e2.last_count_limit() == 960              # (the threshold pinned below the pilot's ratio, ~0.3 at this size: the filter is on)
            got = table(r1)
            assert all(np.array_equal(a, b) for a, b in zip(ref[:3], got[:3])) and ref[3] == got[3]
            r2 = e2.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))      # the hint: no second partition
            assert r2.repartitioned == 0 and r2.n_buckets > 1.1 * nb0 and e2.last_count_limit() == 960
            got2 = table(r2)
            assert all(np.array_equal(a, b) for a, b in zip(ref[:3], got2[:3])) and ref[3] == got2[3]
            tune("SNK_COUNT_SCREEN_NG", "0")
            r3 = e2.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
            assert e2.last_count_limit() == 1920 and r3.n_buckets > 1.5 * nb0
            got3 = table(r3)
            assert all(np.array_equal(a, b) for a, b in zip(ref[:3], got3[:3])) and ref[3] == got3[3]
        finally:
            e2.close()
    finally:
        e.close()
        torch.cuda.empty_cache()


def test_ctx_trim_gives_memory_back_and_the_context_still_works(engine):
    """snk_ctx_trim: the arena's idle blocks go back to the device (another allocator on the GPU can have them), the last result stays
    valid, and the next call allocates again and gives the same answer."""
    import torch
    c = goldens.load("synth_20k_err")
    dev = torch.device("cuda", 0)
    rows = torch.from_numpy(c.rows.view(np.int32).copy()).to(dev)
    quals = torch.from_numpy(np.ascontiguousarray(c.quals)).to(dev)
    bc = torch.from_numpy(c.bc.astype(np.int32)).to(dev)
    lens = torch.from_numpy(c.lens.astype(np.uint16).view(np.int16)).to(dev)
    from supernova_amd.engine import Params
    r1 = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48), ign_bc_below=c.ign_bc_below)
    free0 = torch.cuda.mem_get_info()[0]
    engine.release_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 >= free0
    k1, c1, u1 = r1.keys(), r1.counts(), r1.unitigs()            # still readable after the trim
    r2 = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48), ign_bc_below=c.ign_bc_below)
    assert np.array_equal(k1, r2.keys()) and np.array_equal(c1, r2.counts()) and u1 == r2.unitigs()
    assert np.array_equal(k1[:, :3], c.exp_keys)


def test_ctx_reserve_maps_the_arena_ahead_of_the_calls():
    """snk_ctx_reserve: the arena holds the reserved bytes before the first call, calls of any size leave them mapped (a small call after a
    reservation does not hand it back), the results are the ones of an unreserved context, and snk_ctx_trim lets go of it."""
    import torch
    from supernova_amd.engine import Engine, Params
    c = goldens.load("synth_20k_err")
    rows, quals, bc, lens = _to_dev(c)
    e = Engine(0)
    try:
        free0 = torch.cuda.mem_get_info()[0]
        e.reserve(6 << 30)
        assert free0 - torch.cuda.mem_get_info()[0] >= (6 << 30) - (64 << 20)
        for _ in range(4):
            r = e.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48), ign_bc_below=c.ign_bc_below)
            assert free0 - torch.cuda.mem_get_info()[0] >= (6 << 30) - (64 << 20)         # (small calls do not hand the reservation back)
        _check_against(r, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)
        del r
        e.release_cache()
        assert free0 - torch.cuda.mem_get_info()[0] < 2 << 30
        with pytest.raises(Exception):
            e.reserve(1 << 50)
    finally:
        e.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("name,use_bc", [("adversarial", True), ("synth_20k_err", False)])
def test_k60_vs_oracle(engine, name, use_bc):
    """K=60 (long-k config): key 120 bit, supermers up to 106 bases.  The reference's BuildReadQGraph60 has no barcode
    rule, so K=60 is checked against the C oracle (same algorithm, K generic)."""
    from supernova_amd.engine import Params
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc if use_bc else None, lens=lens, params=Params(K=60),
                             ign_bc_below=c.ign_bc_below)
    gl = oracle_lib.good_lens(c.quals, c.lens, K=60, min_qual=7)
    o = oracle_lib.OracleResult(c.codes, gl, c.bc if use_bc else None, K=60, ign_bc_below=c.ign_bc_below, hbv=False)
    assert np.array_equal(res.good_len().astype(np.uint32), gl)
    assert res.n_instances == o.n_instances
    k = res.keys()
    assert np.array_equal(k, o.keys)
    assert np.array_equal(res.counts(), o.counts)
    assert np.array_equal(res.ctx(), o.ctx)
    assert res.unitigs() == o.unitigs


@pytest.mark.parametrize("long_minimiser", [False, True])
@pytest.mark.parametrize("name", goldens.K60_CASES)
def test_k60_golden(engine, name, long_minimiser):
    """K=60 against golden vectors dumped from the reference's BuildReadQGraph60 (no barcode rule => bc=None)."""
    from supernova_amd.engine import Params
    g = goldens.Case60(name)
    c = g.base
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=None, lens=lens, params=Params(K=60, long_minimiser=long_minimiser))
    assert np.array_equal(res.good_len().astype(np.uint32), g.exp_goodlens)
    assert np.array_equal(res.keys(), g.exp_keys)
    assert np.array_equal(np.minimum(res.counts(), (1 << 24) - 1), g.exp_counts)
    assert np.array_equal(res.ctx(), g.exp_ctx)
    assert res.unitigs() == g.exp_unitigs


def _random_reads(rng, genome_len, n, L, nbc=20, err=0.002):
    g = rng.integers(0, 4, genome_len, dtype=np.uint8)
    codes = np.zeros((n, L), dtype=np.uint8)
    quals = np.full((n, L), 30, dtype=np.uint8)
    lens = np.full(n, L, dtype=np.uint16)
    for i in range(n):
        ln = L if rng.random() < 0.8 else int(rng.integers(20, L + 1))
        s = int(rng.integers(0, genome_len - ln + 1))
        r = g[s:s + ln].copy()
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        e = rng.random(ln) < err
        r[e] = (r[e] + 1) & 3
        quals[i, :ln][e] = 12
        if rng.random() < 0.1:
            quals[i, int(rng.integers(0, ln)):ln] = 2
        codes[i, :ln] = r
        lens[i] = ln
    bc = rng.integers(0, nbc + 1, n).astype(np.int32)
    return codes, quals, lens, bc


@pytest.mark.parametrize("L,K", [(250, 48), (256, 60), (49, 48), (61, 60), (100, 48)])
def test_read_length_extremes_vs_oracle(engine, L, K):
    """Maximum row length (256 bases = 16 words), reads of exactly K+1 bases (2 k-mers), ragged lengths."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Params
    rng = np.random.default_rng(1000 + L + K)
    codes, quals, lens, bc = _random_reads(rng, 3000, 1500 if L > 100 else 6000, L)
    dev = torch.device("cuda", 0)
    rows = torch.from_numpy(synth.pack_rows(codes).view(np.int32)).to(dev)
    res = engine.count_graph(rows, L, quals=torch.from_numpy(quals).to(dev), bc=torch.from_numpy(bc).to(dev),
                             lens=torch.from_numpy(lens.view(np.int16)).to(dev), params=Params(K=K))
    gl = oracle_lib.good_lens(quals, lens, K=K)
    o = oracle_lib.OracleResult(codes, gl, bc, K=K, hbv=False)
    assert np.array_equal(res.good_len().astype(np.uint32), gl)
    assert res.n_instances == o.n_instances
    assert np.array_equal(res.keys(), o.keys)
    assert np.array_equal(res.counts(), o.counts) and np.array_equal(res.ctx(), o.ctx)
    assert res.unitigs() == o.unitigs


@pytest.mark.parametrize("n_buckets", [4, 5, 6])
def test_k60_sparse_chunks_of_a_thousand_fragments(engine, graph_stage, n_buckets):
    """Low coverage, no filter, K=60, ~18 k retained k-mers per bucket: the buckets split 16-32 ways by hash, the
    sub-passes of ~1150 k-mers hold almost only one-k-mer fragments, i.e. more than 65535 fragment bases per chunk -- the
    per-chunk base offsets need 32 bits (found by tests/tools/fuzz_parity.py in a five-rank sharded run; 16-bit offsets
    garbled a few bases of ~90 unitigs, differently in every run)."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Params
    rng = np.random.default_rng(4242)
    codes, quals, lens, bc = _random_reads(rng, 120_000, 2400, 151, err=0.0)
    dev = torch.device("cuda", 0)
    rows = torch.from_numpy(synth.pack_rows(codes).view(np.int32)).to(dev)
    res = engine.count_graph(rows, 151, quals=torch.from_numpy(quals).to(dev), bc=torch.from_numpy(bc).to(dev),
                             lens=torch.from_numpy(lens.view(np.int16)).to(dev),
                             params=Params(K=60, min_freq=1, min_bc=0, n_buckets=n_buckets))
    gl = oracle_lib.good_lens(quals, lens, K=60)
    o = oracle_lib.OracleResult(codes, gl, bc, K=60, min_freq=1, min_bc=0, hbv=False)
    assert res.n_kmers > 60_000 and res.buckets_split >= n_buckets
    if graph_stage == "local":
        assert res.n_fragments > 0.8 * res.n_kmers
    assert np.array_equal(res.keys(), o.keys) and np.array_equal(res.ctx(), o.ctx)
    assert res.unitigs() == o.unitigs


def test_empty_and_degenerate_inputs(engine):
    """No reads; reads that are all too short / all low quality (no k-mer at all); a single read."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Params
    dev = torch.device("cuda", 0)
    L = 150
    for n, qv in [(0, 30), (64, 2), (1, 30), (300, 30)]:
        rng = np.random.default_rng(n + qv)
        codes = rng.integers(0, 4, (n, L), dtype=np.uint8)
        quals = np.full((n, L), qv, dtype=np.uint8)
        lens = np.full(n, 40 if n == 300 else L, dtype=np.uint16)      # 300 reads of 40 bases: shorter than K+1
        rows = torch.from_numpy(synth.pack_rows(codes).view(np.int32).reshape(n, 10)).to(dev)
        res = engine.count_graph(rows, L, quals=torch.from_numpy(quals).to(dev), bc=None,
                                 lens=torch.from_numpy(lens.view(np.int16)).to(dev), params=Params(K=48))
        exp_inst = 103 if (n == 1 and qv == 30) else 0
        assert res.n_instances == exp_inst
        assert res.n_kmers == 0 and res.n_unitigs == 0       # a single read never reaches min_freq = 3
        assert res.keys().shape == (0, 4) and res.unitigs() == []


def test_repeatability_and_full_size_properties(engine):
    """Size-independent properties on a 2 M-read workload (too big for the oracle in the test budget): two runs are
    bit-identical; keys strictly ascending; every count >= min_freq; sum(len-K+1) over unitigs == retained k-mers;
    every unitig is in canonical orientation; spectrum sums to the table size."""
    from supernova_amd import synth
    from supernova_amd.engine import Params
    sp = synth.synth_params(2_000_000, seed=0x5EED0777)
    rows, quals, bc = engine.synth(sp)
    a = engine.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
    ka, ca, xa, ua = a.keys(), a.counts(), a.ctx(), a.unitig_arrays()
    spec_a, nk_a = a.spectrum(), a.n_kmers          # (a result is valid until the next call on its engine)
    b = engine.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, n_buckets=40009))
    assert np.array_equal(ka, b.keys()) and np.array_equal(ca, b.counts()) and np.array_equal(xa, b.ctx())
    ub = b.unitig_arrays()
    assert np.array_equal(ua[0], ub[0]) and np.array_equal(ua[1], ub[1])
    k64 = (ka[:, 0].astype(np.uint64) << np.uint64(32)) | ka[:, 1]
    lo64 = (ka[:, 2].astype(np.uint64) << np.uint64(32)) | ka[:, 3]
    asc = (k64[1:] > k64[:-1]) | ((k64[1:] == k64[:-1]) & (lo64[1:] > lo64[:-1]))
    assert asc.all() and ca.min() >= 3
    off, bases = ua
    lens = (off[1:] - off[:-1]).astype(np.int64)
    assert int((lens - 47).sum()) == nk_a
    assert int(spec_a.sum()) == nk_a
    for i in range(len(lens)):                      # canonical form of every unitig (dna/CanonicalForm.h:35-48)
        s = bases[int(off[i]):int(off[i + 1])]
        if len(s) & 1:
            assert not (s[len(s) // 2] & 2)
        else:
            rc = (3 - s[::-1]).astype(np.uint8)
            j = int(np.argmax(s != rc)) if (s != rc).any() else -1
            assert j < 0 or s[j] < rc[j]


class _DevArr:
    """A device array of the library seen by torch (no copy): __cuda_array_interface__ over the raw pointer."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def test_full_size_properties_1e8(engine, graph_stage):
    """BASELINE config 2 at its full size (100 M x 150 bp, 10.2 G k-mer instances, one GPU), checked on the device through
    size-independent properties: strictly ascending keys, every count >= min_freq, spectrum and unitig lengths add up
    to the table size, and -- the checksum of checksums -- table and unitigs are identical when the same reads go
    through a different bucket count (different supermer grouping, different chunking, different fragments)."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Params
    if graph_stage == "global":
        pytest.skip("one pass over the full size is enough; the global stage is cross-checked on the small cases")
    n = 100_000_000
    sp = synth.synth_params(n, seed=0x5EED0002)
    rows, quals, bc = engine.synth(sp)

    def run(nb):
        r = engine.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, n_buckets=nb))
        nk = r.n_kmers
        keys = torch.as_tensor(_DevArr(r.raw.keys, 2 * nk, "<i8"), device="cuda").view(nk, 2)      # lo, hi
        cnt = torch.as_tensor(_DevArr(r.raw.counts, nk, "<i4"), device="cuda")
        ctx = torch.as_tensor(_DevArr(r.raw.ctx, nk, "|u1"), device="cuda")
        lo, hi = keys[:, 0], keys[:, 1]
        # ascending as unsigned 128-bit numbers: flip the sign bit to compare int64 as uint64
        f = lambda t: t ^ torch.tensor(-(1 << 63), dtype=torch.int64, device="cuda")
        h0, h1, l0, l1 = f(hi[:-1]), f(hi[1:]), f(lo[:-1]), f(lo[1:])
        assert bool(((h1 > h0) | ((h1 == h0) & (l1 > l0))).all())
        assert int(cnt.min()) >= 3
        chk = (hi * 0x9E3779B97F4A7C15 + lo * 0x42B2AE3D27D4EB4F + cnt.to(torch.int64) * 0x165667B19E3779F9
               + ctx.to(torch.int64) * 0x27D4EB2F165667C5).sum()
        spec = torch.as_tensor(_DevArr(r.raw.spectrum, int(r.raw.spectrum_bins), "<i8"), device="cuda")
        assert int(spec.sum()) == nk
        off = torch.as_tensor(_DevArr(r.raw.unitig_off, r.n_unitigs + 1, "<i8"), device="cuda")
        bases = torch.as_tensor(_DevArr(r.raw.unitig_bases, r.unitig_total_bases, "|u1"), device="cuda")
        assert int((off[1:] - off[:-1] - 47).sum()) == nk
        return dict(n_inst=r.n_instances, nk=nk, chk=int(chk), nu=r.n_unitigs, off=off.clone(), bases=bases.clone())

    a = run(0)
    assert a["n_inst"] > 10_000_000_000 and a["nk"] > 200_000_000
    b = run(1_500_007)
    assert (a["n_inst"], a["nk"], a["chk"], a["nu"]) == (b["n_inst"], b["nk"], b["chk"], b["nu"])
    assert torch.equal(a["off"], b["off"]) and torch.equal(a["bases"], b["bases"])


def test_full_size_booked_slots_1e8(engine, graph_stage, monkeypatch, tune):
    """100 M reads with 1.5 % substitutions and long low-quality tails (config.robust's second model) at full size: the call that lets the
    data switch the count kernel to the bit filter + booked slots (snk_ctx_last_count_limit = 960; 2.6 x the distinct k-mers of clean reads),
    the one with booked slots alone (1920) and the default kernel on smaller buckets (1216) give the same table -- checksum over keys, counts
    and contexts -- and the same unitigs."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params
    if graph_stage == "global":
        pytest.skip("one pass over the full size is enough")
    n = 100_000_000
    sp = synth.synth_params(n, seed=0x5EED0042, sub_ppm=15000, lowq_tail_ppm=500000)
    e = Engine(0)
    try:
        rows, quals, bc = e.synth(sp)

        def run():
            r = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
            nk = r.n_kmers
            keys = torch.as_tensor(_DevArr(r.raw.keys, 2 * nk, "<i8"), device="cuda").view(nk, 2)
            cnt = torch.as_tensor(_DevArr(r.raw.counts, nk, "<i4"), device="cuda")
            ctx = torch.as_tensor(_DevArr(r.raw.ctx, nk, "|u1"), device="cuda")
            chk = (keys[:, 1] * 0x9E3779B97F4A7C15 + keys[:, 0] * 0x42B2AE3D27D4EB4F + cnt.to(torch.int64) * 0x165667B19E3779F9
                   + ctx.to(torch.int64) * 0x27D4EB2F165667C5).sum()
            off = torch.as_tensor(_DevArr(r.raw.unitig_off, r.n_unitigs + 1, "<i8"), device="cuda")
            bases = torch.as_tensor(_DevArr(r.raw.unitig_bases, r.unitig_total_bases, "|u1"), device="cuda")
            return dict(n_inst=r.n_instances, nk=nk, chk=int(chk), nu=r.n_unitigs, nb=r.n_buckets, off=off.clone(), bases=bases.clone(), lim=e.last_count_limit())

        run()                       # (the first call looks at the first buckets and partitions again)
        a = run()                   # 0.4 distinct k-mers per instance: the bit filter in front of a 1024-slot table
        assert a["lim"] == 960 and a["nk"] > 200_000_000
        tune("SNK_COUNT_SCREEN_NG", "0")
        c = run()                   # booked slots alone
        assert c["lim"] == 1920 and c["nb"] > a["nb"]
        tune("SNK_COUNT_TIGHT", "0")
        b = run()                   # the default kernel on still smaller buckets
        assert b["lim"] == 1216 and b["nb"] > c["nb"]
        for o in (b, c):
            assert (a["n_inst"], a["nk"], a["chk"], a["nu"]) == (o["n_inst"], o["nk"], o["chk"], o["nu"])
            assert torch.equal(a["off"], o["off"]) and torch.equal(a["bases"], o["bases"])
    finally:
        e.close()
        torch.cuda.empty_cache()


def test_unsorted_table_mode(engine, graph_stage):
    """SNK_F_UNSORTED_TABLE: same table (as a set) and the same unitigs, keys left in bucket order."""
    from supernova_amd.engine import Params
    c = goldens.load("adversarial")
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48, sorted_table=False),
                             ign_bc_below=c.ign_bc_below)
    k, cnt, ctx = res.keys(), res.counts(), res.ctx()
    order = np.lexsort((k[:, 2], k[:, 1], k[:, 0]))
    assert np.array_equal(k[order][:, :3], c.exp_keys)
    assert np.array_equal(np.minimum(cnt[order], (1 << 24) - 1), c.exp_counts)
    assert np.array_equal(ctx[order], c.exp_ctx)
    assert res.unitigs() == c.exp_unitigs
    if graph_stage == "local":
        assert res.n_fragments >= res.n_unitigs and res.n_boundary > 0


def test_unitig_order_is_deterministic(engine, graph_stage):
    """Unitigs leave the device ordered by their first K bases, independent of workgroup scheduling."""
    c = goldens.load("synth_20k_err")
    rows, quals, bc, lens = _to_dev(c)
    outs = []
    for _ in range(3):
        res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
        off, bases = res.unitig_arrays()
        outs.append((off.copy(), bases.copy()))
    for o, b in outs[1:]:
        assert np.array_equal(o, outs[0][0]) and np.array_equal(b, outs[0][1])
    off, bases = outs[0]
    firsts = [bytes(bases[int(off[i]):int(off[i]) + 48]) for i in range(len(off) - 1)]
    if graph_stage == "local":
        assert firsts == sorted(firsts)


@pytest.mark.parametrize("pct", [100, 60, 10])
def test_partition_overflow_segment(engine, monkeypatch, pct, tune):
    """The single-pass minimiser partition gives every bucket a fixed number of record slots; supermers beyond it go
    through the overflow list (second count segment).  Shrinking the capacity must not change any result."""
    tune("SNK_MSP_CAP_PCT", str(pct))
    c = goldens.load("synth_20k_err")
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)
    if pct < 100:
        assert res.n_overflow > 0


@pytest.mark.parametrize("name", ["adversarial", "synth_20k_err"])
@pytest.mark.parametrize("hot_buckets", [False, True])
def test_partition_stops_reserving_slots_for_hot_buckets(engine, monkeypatch, name, hot_buckets, tune):
    """snk_msp.hip: a bucket that was handed a slot far beyond its capacity is noted in a small table; workgroups that start later send its
    supermers to the overflow list without touching its cursor (same-address atomics queue: a homopolymer's bucket held the partition
    for 60 ms).  Forced on the goldens: tiny capacity, noted at the first overflowing slot -- same results, with and without the
    k-mer-hash re-partition of the buckets that end up hot."""
    tune("SNK_MSP_CAP_PCT", "10")
    tune("SNK_MSP_HOT_FACTOR", "1")
    tune("SNK_MSP_HOT_MIN", "1")
    if hot_buckets:
        tune("SNK_HOT_MIN", "8")
        tune("SNK_HOT_FACTOR", "1")
        tune("SNK_HOT_CLASS_INST", "300")
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)
    assert res.n_overflow > 0
    assert res.n_supermers == engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below).n_supermers
    tune("SNK_MSP_HOT_FACTOR", "0")          # never noted: every supermer takes its reservation
    assert res.n_supermers == engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below).n_supermers


@pytest.mark.parametrize("name,K", [("adversarial", 48), ("synth_20k_err", 48), ("adversarial", 60)])
def test_dense_partition_mode(engine, monkeypatch, name, K, tune):
    """SNK_MSP_DENSE=1 (round 4's measured alternative to the slot reservations, DESIGN 4 "round 4"): records leave the scan kernel in
    read order without any per-bucket atomic, (bucket, position) pairs are radix-sorted, the count kernel gathers a bucket's records
    through the sorted positions.  Same results bit for bit; there is no overflow segment in this mode."""
    from supernova_amd.engine import Params
    tune("SNK_MSP_DENSE", "1")
    if K == 48:
        c = goldens.load(name)
        rows, quals, bc, lens = _to_dev(c)
        res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48), ign_bc_below=c.ign_bc_below)
        _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)
    else:
        g = goldens.Case60(name)
        c = g.base
        rows, quals, bc, lens = _to_dev(c)
        res = engine.count_graph(rows, c.read_len, quals=quals, bc=None, lens=lens, params=Params(K=60))
        assert np.array_equal(res.keys(), g.exp_keys) and np.array_equal(np.minimum(res.counts(), (1 << 24) - 1), g.exp_counts)
        assert np.array_equal(res.ctx(), g.exp_ctx) and res.unitigs() == g.exp_unitigs
    assert res.n_overflow == 0 and res.n_supermers > 0


@pytest.mark.parametrize("passes", [1, 3])
@pytest.mark.parametrize("name,K", [("adversarial", 48), ("synth_20k_err", 48), ("adversarial", 60), ("synth_20k_err", 60)])
def test_hot_buckets_are_repartitioned_by_kmer_hash(engine, monkeypatch, name, K, passes, tune):
    """snk_hot.hip: a minimiser bucket far above its capacity (a repeat family, a homopolymer run) is expanded into single-k-mer records
    that go to virtual buckets (bucket, hash class), counted by other workgroups in a second launch of the count kernel.  Forced here
    on the goldens by a tiny capacity and a tiny threshold (every overflowing bucket is 'hot', classes of ~300 instances so that they
    split further); same results bit for bit."""
    from supernova_amd.engine import Params
    tune("SNK_MSP_CAP_PCT", "10")
    tune("SNK_HOT_MIN", "8")
    tune("SNK_HOT_FACTOR", "1")
    tune("SNK_HOT_CLASS_INST", "300")
    if passes > 1:          # bucket-range passes: the hot buckets of a range are expanded while that range's records are in the slot array
        tune("SNK_PARTITION_PASSES", str(passes))
    if K == 48:
        c = goldens.load(name)
        rows, quals, bc, lens = _to_dev(c)
        res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48), ign_bc_below=c.ign_bc_below)
        _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)
    else:
        g = goldens.Case60(name)
        c = g.base
        rows, quals, bc, lens = _to_dev(c)
        res = engine.count_graph(rows, c.read_len, quals=quals, bc=None, lens=lens, params=Params(K=60))
        assert np.array_equal(res.keys(), g.exp_keys) and np.array_equal(np.minimum(res.counts(), (1 << 24) - 1), g.exp_counts)
        assert np.array_equal(res.ctx(), g.exp_ctx) and res.unitigs() == g.exp_unitigs
    assert res.n_hot_buckets > 0 and res.n_overflow > 0 and engine.last_partition_passes() == passes


@pytest.mark.parametrize("persist,n_buckets", [(0, 0), (1, 0), (32, 3), (32, 5000), (1000, 5000)])
def test_count_launch_shapes(engine, monkeypatch, persist, n_buckets, tune):
    """The count kernel's workgroups walk strided bucket lists (SNK_COUNT_PERSIST residency waves; 0 = one bucket per
    workgroup): fewer buckets than workgroups, one wave, many waves, more waves than buckets -- same results, including
    the split path (3 buckets for 20 k reads overflow the LDS table)."""
    tune("SNK_COUNT_PERSIST", str(persist))
    from supernova_amd.engine import Params
    c = goldens.load("synth_20k_err")
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48, n_buckets=n_buckets),
                             ign_bc_below=c.ign_bc_below)
    if n_buckets == 3:
        assert res.buckets_split >= 1
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)


@pytest.mark.parametrize("min_freq,n_buckets,screen", [(2, 0, "2"), (2, 0, "1"), (3, 0, "1"), (1, 0, "1"), (5, 0, "1"), (3, 40, "1"), (3, 3000, "1"), (3, 0, "0")])
def test_grouped_per_barcode_graphs(engine, graph_stage, monkeypatch, min_freq, n_buckets, screen, tune):
    """BASELINE config 5: per-group (per-barcode) local graphs.  One grouped run == the oracle applied to every group's
    reads on its own (frequency rule only): tables, pruned contexts and unitigs per group.  The count kernel's bit filter in front of the
    table (snk_count.hip SCREEN: levels 2 and 3; off at min_freq 1; buckets too large for it -- 40 buckets here -- are counted without it)
    does not change the result."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Params
    if graph_stage == "global":
        pytest.skip("grouped runs use the bucket-local stage")
    c = goldens.load("synth_20k_err")
    rng = np.random.default_rng(7)
    n = c.rows.shape[0]
    NG = 5
    group = (c.bc.astype(np.int64) % NG).astype(np.int32)
    group[rng.random(n) < 0.1] = NG + 3                     # a sparse extra group with a large id gap
    rows, quals, bc, lens = _to_dev(c)
    g_dev = torch.from_numpy(group).to(rows.device)
    tune("SNK_COUNT_SCREEN", screen)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=None, lens=lens, group=g_dev,
                             params=Params(K=48, min_freq=min_freq, min_bc=0, grouped=True, sorted_table=False, n_buckets=n_buckets))
    k, cnt, ctx = res.keys(), res.counts(), res.ctx()
    off, bases = res.unitig_arrays()
    ug = res.unitig_groups()
    assert np.all(np.diff(ug.astype(np.int64)) >= 0)                      # group-major order
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    total = 0
    for gid in np.unique(group):
        sel = group == gid
        o = oracle_lib.OracleResult(c.codes[sel], c.exp_goodlens[sel], None, min_freq=min_freq, min_bc=0, hbv=False)
        m = k[:, 3] == gid
        kk, cc, xx = k[m], cnt[m], ctx[m]
        order = np.lexsort((kk[:, 2], kk[:, 1], kk[:, 0]))
        assert np.array_equal(kk[order][:, :3], o.keys[:, :3]), gid
        assert np.array_equal(cc[order], o.counts), gid
        assert np.array_equal(xx[order], o.ctx), gid
        us = sorted((lut[bases[int(off[u]):int(off[u + 1])]].tobytes().decode() for u in np.nonzero(ug == gid)[0]),
                    key=lambda t: (-len(t), t))
        assert us == o.unitigs, gid
        total += m.sum()
    assert total == k.shape[0]


def _repeat_rich_reads(n_reads, seed):
    """Low-complexity torture at scale: dispersed repeats, poly-A runs, short-period tandem repeats, planted palindromes.
    Exercises what random sequence never does: supermer lists that overflow, buckets that split, chunks above 256 k-mers,
    circles inside one bucket."""
    rng = np.random.default_rng(seed)
    G = rng.integers(0, 4, 150_000).astype(np.uint8)
    elem = rng.integers(0, 4, 300).astype(np.uint8)
    for p0 in rng.integers(0, len(G) - 400, 40):
        G[p0:p0 + 300] = elem
    for p0 in rng.integers(0, len(G) - 400, 25):
        G[p0:p0 + int(rng.integers(60, 220))] = 0                       # poly-A
    for p0 in rng.integers(0, len(G) - 400, 40):
        per = int(rng.integers(1, 13))
        unit = rng.integers(0, 4, per).astype(np.uint8)
        ln = int(rng.integers(100, 320))
        G[p0:p0 + ln] = np.resize(unit, ln)                             # tandem repeat
    for p0 in rng.integers(0, len(G) - 400, 6):
        h = rng.integers(0, 4, 24).astype(np.uint8)
        G[p0:p0 + 48] = np.concatenate([h, (3 - h)[::-1]])              # reverse-complement palindrome (a 1-k-mer unitig)
    L = 150
    st = rng.integers(0, len(G) - L, n_reads)
    codes = G[st[:, None] + np.arange(L)[None, :]]
    flip = rng.random(n_reads) < 0.5
    codes[flip] = (3 - codes[flip])[:, ::-1]
    err = rng.random(codes.shape) < 0.003
    codes = np.where(err, (codes + rng.integers(1, 4, codes.shape)) % 4, codes).astype(np.uint8)
    quals = np.where(err, 12, 30).astype(np.uint8)
    tails = rng.random(n_reads) < 0.05
    tl = rng.integers(1, 50, n_reads)
    quals[tails[:, None] & (np.arange(L)[None, :] >= (L - tl)[:, None])] = 2
    bc = rng.integers(0, 40, n_reads).astype(np.int32)
    return np.ascontiguousarray(codes), quals, bc


def test_repeat_rich_vs_oracle(engine):
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Params
    codes, quals, bc = _repeat_rich_reads(60_000, 99)
    dev = torch.device("cuda", 0)
    rows = torch.from_numpy(synth.pack_rows(codes).view(np.int32)).to(dev)
    res = engine.count_graph(rows, 150, quals=torch.from_numpy(quals).to(dev), bc=torch.from_numpy(bc).to(dev),
                             params=Params(K=48, n_buckets=61))          # few buckets: splits and big chunks
    gl = oracle_lib.good_lens(quals, 150)
    o = oracle_lib.OracleResult(codes, gl, bc, hbv=False)
    hist = np.bincount(np.minimum(o.counts, (1 << 24) - 1)).astype(np.int64)
    _check_against(res, o.keys[:, :3], np.minimum(o.counts, (1 << 24) - 1), o.ctx, o.unitigs, gl, hist)
    assert res.buckets_split > 0
    res2 = engine.count_graph(rows, 150, quals=torch.from_numpy(quals).to(dev), bc=torch.from_numpy(bc).to(dev), params=Params(K=48))
    _check_against(res2, o.keys[:, :3], np.minimum(o.counts, (1 << 24) - 1), o.ctx, o.unitigs, gl, hist)


def test_vs_reference_binary_200k(engine, graph_stage, tmp_path):
    """The reference ITSELF (oracle/_ref/snref_driver, built from /root/reference by oracle/ref/build_ref.sh and carried to
    the GPU box) against the HIP path on a fresh seeded 200 k-read workload of the bench's model -- no committed fixture in
    between: good lengths, retained table, contexts, spectrum, unitigs, bit for bit."""
    import os
    import torch
    import refio
    from supernova_amd import synth
    if graph_stage == "global":
        pytest.skip("one stage is enough for this one")
    if not refio.REF_DRIVER.exists():
        pytest.skip("oracle/_ref/snref_driver not built (needs /root/reference in the build container)")
    n = 200_000
    sp = synth.synth_params(n, seed=0x5EED0777)
    rows, quals, bc = synth.synth_host(sp)
    asc = synth.codes_to_ascii(synth.unpack_rows(rows, 150))
    refio.write_snkrd(tmp_path / "in.snkrd", np.full(n, 150), asc, quals, bc)
    refio.run_ref(tmp_path / "in.snkrd", tmp_path / "out", threads=min(32, os.cpu_count() or 8))
    d = refio.read_ref_dump(tmp_path / "out")
    dev = torch.device("cuda", 0)
    res = engine.count_graph(torch.from_numpy(rows.view(np.int32)).to(dev), 150, quals=torch.from_numpy(quals).to(dev),
                             bc=torch.from_numpy(bc).to(dev))
    hist = np.asarray(d["hist"]["vals"], dtype=np.int64)
    _check_against(res, d["kmers"]["k"], d["kmers"]["count"], d["kmers"]["ctx"], d["unitigs"], d["goodlens"], hist)
    # f1: the same run's read paths (the reference's pathReads, new aligner) against the device pather
    rows_d, quals_d = torch.from_numpy(rows.view(np.int32)).to(dev), torch.from_numpy(quals).to(dev)
    off, ne, edges, info = res.path_reads(rows_d, 150, quals_d, mark_dups=True, bc=torch.from_numpy(bc).to(dev))
    assert np.array_equal(ne.astype(np.int32), d["path_n"]) and np.array_equal(edges, d["path_edges"]) and np.array_equal(off, d["path_off"])
    assert int((ne > 1).sum()) > 100 and int((ne == 0).sum()) > 0          # multi-edge paths and unplaced reads both occur
    # f4: the same run's MarkDups (100 k pairs on a 0.5 Mb genome: chance duplicates by the hundred)
    assert np.array_equal(info["dups"]["dup"], d["dup"]) and info["dups"]["interdup_rate"] == d["interdup"]
    assert int(d["dup"].sum()) > 50


@pytest.mark.parametrize("name", ["adversarial", "synth_20k_err"])
def test_device_bv_image(engine, graph_stage, tmp_path, name):
    """a13 on the device (snk_dev_bv_image): BVComp order (length descending, then lexicographic: HBVFromEdges.cc:106-111), 2-bit packing
    and the header in HBM -- the bytes of the file the host writer (and the oracle's independent writer, tests/test_graphio.py) produce
    from the reference's unitigs put into that order."""
    from supernova_amd import graphio
    if graph_stage == "global":
        pytest.skip("one stage is enough for this one")
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
    img = res.bv_image()
    us = sorted(c.exp_unitigs, key=lambda u: (-len(u), u))
    off, bases = graphio.unitigs_to_arrays(us)
    p = tmp_path / "host.bv"
    graphio.write_bv(p, off, bases)
    assert img == p.read_bytes()


@pytest.mark.parametrize("name,cuts", [("adversarial", (0.0, 0.31, 0.32, 0.9, 1.0)), ("synth_20k_err", (0.0, 0.5, 1.0)), ("synth_2k_err", (0.0, 1.0))])
def test_streamed_slabs_equal_the_resident_call(engine, graph_stage, name, cuts):
    """snk_dev_stream_begin / _append / _finish: the reads arrive in slabs (ragged cuts, an empty slab, slabs with and without the fused
    trim's alignment), every slab is partitioned as it arrives and may be freed afterwards -- the reads are never resident as a whole.
    Result == the golden (= the resident call) bit for bit; reads beyond the job's bound and calls without a job are refused."""
    import torch
    from supernova_amd.engine import Params
    from supernova_amd.lib import SnkError
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    if name != "adversarial":          # quality rows padded to a multiple of four bytes: the trim runs inside the partition kernel
        quals = torch.nn.functional.pad(quals, (0, 160 - quals.shape[1])).contiguous()
    n = rows.shape[0]
    bounds = [int(round(f * n)) & ~1 for f in cuts]
    bounds[-1] = n
    engine.stream_begin(c.read_len, n + 10, has_bc=True, params=Params(K=48))
    for a, b in zip(bounds[:-1], bounds[1:]):
        # every slab in memory of its own that is dropped right after the call (stream order keeps it alive long enough: torch's caching
        # allocator does not hand a freed block to another stream)
        r, q, bcs, ln = rows[a:b].clone(), quals[a:b].clone(), bc[a:b].clone(), lens[a:b].clone()
        engine.stream_append(r, c.read_len, quals=q, bc=bcs, lens=ln, ign_bc_below=c.ign_bc_below, read_index_base=a)
        torch.cuda.synchronize()
        del r, q, bcs, ln
    engine.stream_append(rows[:0].clone(), c.read_len, quals=quals[:0].clone(), bc=bc[:0].clone(), lens=lens[:0].clone())       # an empty slab
    res = engine.stream_finish()
    assert res.n_reads == n
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)
    with pytest.raises(SnkError):
        engine.stream_finish()                                  # no open job
    engine.stream_begin(c.read_len, n // 2, has_bc=True, params=Params(K=48))
    with pytest.raises(SnkError, match="upper bound"):
        engine.stream_append(rows, c.read_len, quals=quals, bc=bc, lens=lens)


@pytest.mark.parametrize("passes", [2, 3, 7])
@pytest.mark.parametrize("name,K", [("adversarial", 48), ("synth_20k_err", 48), ("synth_2k_err", 48)] + [(n, 60) for n in goldens.K60_CASES[:1]])
def test_bucket_range_passes_equal_the_one_pass_partition(engine, graph_stage, name, K, passes, monkeypatch, tune):
    """A job whose supermer slots would not fit the device is partitioned and counted in bucket-range passes over ONE slot array, the
    reads scanned once per pass (snk_partition_passes; the reference re-scans in passes when its records do not fit,
    MapReduceEngine.h:452-468, utils.rs:329-341).  Forced here at golden size: the result is the golden bit for bit, with the quality
    trim inside the first pass (rows padded to four bytes) and with the separate trim kernel."""
    import torch
    from supernova_amd.engine import Params
    tune("SNK_PARTITION_PASSES", str(passes))
    g = goldens.Case60(name) if K == 60 else None
    c = g.base if g else goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    if name != "adversarial":
        quals = torch.nn.functional.pad(quals, (0, 160 - quals.shape[1])).contiguous()
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc if K == 48 else None, lens=lens, params=Params(K=K), ign_bc_below=c.ign_bc_below)
    assert engine.last_partition_passes() == passes
    if g:
        assert np.array_equal(res.good_len().astype(np.uint32), g.exp_goodlens) and np.array_equal(res.keys(), g.exp_keys)
        assert np.array_equal(np.minimum(res.counts(), (1 << 24) - 1), g.exp_counts) and np.array_equal(res.ctx(), g.exp_ctx)
        assert res.unitigs() == g.exp_unitigs
    else:
        _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)
    engine.clear_option("partition_passes")
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc if K == 48 else None, lens=lens, params=Params(K=K), ign_bc_below=c.ign_bc_below)
    assert engine.last_partition_passes() == 1


@pytest.mark.parametrize("coverage", [56, 28])
def test_a_small_device_plans_passes_and_probes_its_regions(engine, graph_stage, coverage, tune):
    """What a job that approaches the device's memory goes through (800 M reads on one GPU: tools/r6_full_job.py), at 2 M reads on a context that
    is told it has 256 MB (option plan_mem_mb): bucket-range passes by the library's own plan, 3-sigma slots, count regions from
    instances / 32 -- enough at 56x coverage, too small at 28x (1 in 19 retained), where the first range's probe makes them larger and that
    range runs again -- and the record slots handed back before the dense table.  Same table, contexts and unitigs as the one-pass call."""
    from supernova_amd import synth
    from supernova_amd.engine import Params
    if graph_stage == "global":
        pytest.skip("the plans are the count stage's; one graph stage is enough")
    n = 2_000_000
    sp = synth.synth_params(n, seed=0x5EED0A11, genome_len=n * 150 // coverage)
    rows, quals, bc = engine.synth(sp)

    def sig(res):
        return res.n_instances, res.keys().tobytes(), res.counts().tobytes(), res.ctx().tobytes(), res.unitigs()

    one = engine.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
    assert engine.last_partition_passes() == 1
    a = sig(one)
    assert one.n_kmers > (n * 103 // 32 if coverage == 28 else 0)        # (28x: more survivors than the tight plan's first guess)
    tune("plan_mem_mb", 256)
    for _ in range(2):                  # (the second call sizes its regions from the first)
        res = engine.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
        assert engine.last_partition_passes() > 1
        assert sig(res) == a


def test_a_small_device_lets_its_first_range_choose_the_count_kernel(engine, graph_stage, tune):
    """Error-rich reads (1.5 % substitutions) on a NEW context that is told it has 256 MB: the job runs in bucket-range passes, the first
    buckets of its first range are the pilot -- their tables run full, the job is partitioned again into smaller buckets and counted behind
    the bit filter (960 usable slots), as the one-pass path does -- and the result is the one-pass call's."""
    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params
    if graph_stage == "global":
        pytest.skip("the plans are the count stage's; one graph stage is enough")
    n = 2_000_000
    sp = synth.synth_params(n, seed=0x5EED0A12, sub_ppm=15000, lowq_tail_ppm=500000)
    rows, quals, bc = engine.synth(sp)
    one = engine.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
    a = (one.n_instances, one.keys().tobytes(), one.counts().tobytes(), one.ctx().tobytes(), one.unitigs())
    tune("plan_mem_mb", 256)
    e2 = Engine(0)
    try:
        res = e2.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
        assert e2.last_partition_passes() > 1 and res.repartitioned == 1 and e2.last_count_limit() in (960, 1920)
        assert (res.n_instances, res.keys().tobytes(), res.counts().tobytes(), res.ctx().tobytes(), res.unitigs()) == a
    finally:
        e2.close()


def test_open_streamed_job_dies_with_its_arena(engine):
    """A streamed job keeps its slots, cursors and good lengths in the context's arena.  A resident call in between recycles that arena
    (snk_ctx_release_scratch): append / finish must then be refused instead of writing into memory that belongs to the new call
    (snk_pipeline.hip)."""
    from supernova_amd.engine import Params
    from supernova_amd.lib import SnkError
    c = goldens.load("synth_2k_err")
    rows, quals, bc, lens = _to_dev(c)
    n = rows.shape[0]
    engine.stream_begin(c.read_len, n, has_bc=True, params=Params(K=48))
    engine.stream_append(rows[: n // 2].clone(), c.read_len, quals=quals[: n // 2].clone(), bc=bc[: n // 2].clone(), lens=lens[: n // 2].clone())
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)     # takes the arena
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)
    with pytest.raises(SnkError, match="no open job"):
        engine.stream_append(rows[n // 2:].clone(), c.read_len, quals=quals[n // 2:].clone(), bc=bc[n // 2:].clone(), lens=lens[n // 2:].clone())
    with pytest.raises(SnkError, match="no open job"):
        engine.stream_finish()


def test_circle_pool_retry(engine, monkeypatch, tune):
    """Circles inside one chunk take their fragment slots from a small pool; an empty pool must trigger the exact re-run."""
    tune("SNK_BL_POOL", "0")
    c = goldens.load("adversarial")
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
    _check_against(res, c.exp_keys, c.exp_counts, c.exp_ctx, c.exp_unitigs, c.exp_goodlens, c.exp_hist)


HBV_FLOODS = {"host": {"SNK_HBV_DEV_MIN": "1000000000"},            # the sequential flood over the downloaded classes
              "device": {"SNK_HBV_DEV_MIN": "0"},                    # components on the device, one thread floods one component
              "device_big4": {"SNK_HBV_DEV_MIN": "0", "SNK_HBV_BIG": "4"},   # components above 4 nodes go to the host's flood
              "device_big0": {"SNK_HBV_DEV_MIN": "0", "SNK_HBV_BIG": "0"}}   # every component does


@pytest.mark.parametrize("flood", list(HBV_FLOODS))
@pytest.mark.parametrize("name,K", [(n, 48) for n in goldens.CASES] + [(n, 60) for n in goldens.K60_CASES])
def test_device_hbv_matches_reference(engine, graph_stage, name, K, flood, monkeypatch, tune):
    """a14 on the device (snk_dev_hbv): BVComp ranking, (K-1)-mer end keys, vertex classes, connected components and the id
    flood per component in HBM (or, `host`, the flood on the host) -- the text dump equals the reference's buildHBVFromEdges
    output (golden), and the host-array entry point."""
    from supernova_amd import graphio
    from supernova_amd.engine import Params
    if graph_stage == "global" and flood != "device":
        pytest.skip("the flood does not depend on the graph stage")
    for k, v in HBV_FLOODS[flood].items():
        tune(k, v)
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    exp = c if K == 48 else goldens.Case60(name)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc if K == 48 else None, lens=lens, params=Params(K=K),
                             ign_bc_below=c.ign_bc_below)
    off, bases = res.unitig_arrays()
    h = res.hbv()
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[bases].tobytes().decode()
    dev_order = [asc[int(off[i]):int(off[i + 1])] for i in range(res.n_unitigs)]
    ranked = [dev_order[i] for i in h["order"]]
    assert ranked == exp.exp_unitigs                     # BVComp order, computed on the device
    assert graphio.hbv_text(ranked, h) == exp.exp_hbv
    off2, bases2 = graphio.unitigs_to_arrays(ranked)
    h2 = graphio.hbv_from_unitigs(K, off2, bases2)
    for k in ("v_left", "v_right", "src", "is_rc", "fwd", "rev"):
        assert np.array_equal(h[k], h2[k]), k
    # a second call on the same result (scratch of the first was handed back, the unitigs were not)
    assert np.array_equal(res.hbv()["v_left"], h["v_left"])


@pytest.mark.parametrize("flood", ["host", "device", "device_big4"])
def test_device_hbv_many_unitigs(engine, flood, monkeypatch, tune):
    """Error-rich reads at low coverage: hundreds of thousands of short unitigs, equal lengths everywhere (the BVComp
    tie-break) -- device result == host-array entry point on the BVComp-sorted unitigs."""
    from supernova_amd import graphio, synth
    from supernova_amd.engine import Params
    for k, v in HBV_FLOODS[flood].items():
        tune(k, v)
    sp = synth.synth_params(400_000, seed=0x5EED0042)
    rows, quals, bc = engine.synth(sp)
    res = engine.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, min_freq=1, min_bc=0))
    assert res.n_unitigs > 50_000
    off, bases = res.unitig_arrays()
    h = res.hbv()
    us = res.unitigs()
    lens = np.diff(off).astype(np.int64)
    ranked_len = lens[h["order"]]
    assert np.all(np.diff(ranked_len) <= 0)
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[bases].tobytes().decode()
    ranked = [asc[int(off[i]):int(off[i + 1])] for i in h["order"]]
    assert ranked == us
    off2, bases2 = graphio.unitigs_to_arrays(us)
    h2 = graphio.hbv_from_unitigs(48, off2, bases2)
    assert h["n_vertices"] == h2["n_vertices"] and h["n_edges"] == h2["n_edges"]
    for k in ("v_left", "v_right", "src", "is_rc", "fwd", "rev"):
        assert np.array_equal(h[k], h2[k]), k


@pytest.mark.parametrize("lookup", ["index", "kmer_dictionary"])
@pytest.mark.parametrize("name", goldens.CASES)
def test_read_paths_match_reference(engine, graph_stage, name, lookup, monkeypatch, tune):
    """f1: read pathing on the device (look-ups through the minimiser index over the unitigs -- or, SNK_PATH_INDEX=0, the k-mer
    dictionary of rounds 1-3 --, wave-per-read seed and extend, algorithmTwo, quality-aware extension) against the paths the
    reference binary dumped: offset and HBV edge ids of every read, bit for bit."""
    if graph_stage == "global":
        pytest.skip("pathing reads the unitigs; one graph stage is enough")
    if lookup == "index":       # the k-mer dictionary "does not fit": the call falls back to the index by itself
        tune("SNK_PATH_DICT_MAX_KB", "1")
    else:
        tune("SNK_PATH_INDEX", "0")
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
    off, ne, edges, info = res.path_reads(rows, c.read_len, quals, lens=lens)
    assert info["lookup"] == lookup
    bad = np.nonzero(ne.astype(np.int64) != c.exp_path_n)[0]
    assert len(bad) == 0, (len(bad), bad[:5], ne[bad[:5]], c.exp_path_n[bad[:5]])
    assert np.array_equal(edges, c.exp_path_edges)
    assert np.array_equal(off, c.exp_path_off)


@pytest.mark.parametrize("name", goldens.CASES)
def test_mark_dups_match_reference(engine, graph_stage, name):
    """f4: MarkDups on the device (radix sort of (first edge, offset, mate head), one thread per duplicate group) over the
    device's own read paths, against the reference's MarkDups on the golden cases: the flag of every pair, the inter-barcode
    rate exactly, the artifactual pairs against the C restatement (the reference only logs their percentage)."""
    if graph_stage == "global":
        pytest.skip("duplicates are marked on read paths; one graph stage is enough")
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
    off, ne, edges, info = res.path_reads(rows, c.read_len, quals, lens=lens, mark_dups=True, bc=bc)
    d = info["dups"]
    assert np.array_equal(d["dup"], c.exp_dup), np.nonzero(d["dup"] != c.exp_dup)[0][:10]
    assert d["interdup_rate"] == c.exp_interdup
    o_dup, o_art, o_rate, o_nd, o_ni = oracle_lib.mark_dups(c.codes, c.quals, c.lens, c.exp_path_off, c.exp_path_n, c.exp_path_edges, bc=c.bc)
    assert (d["n_dup_reads"], d["n_interdup_reads"], d["n_dup_pairs"], d["n_art_pairs"]) == (o_nd, o_ni, int(o_dup.sum()), int(o_art.sum()))
    assert d["n_placed"] == int((c.exp_path_n > 0).sum())
    assert float(f"{100.0 * d['n_art_pairs'] / len(o_dup):.2g}") == float(f"{c.exp_art_perc:.2g}")


def test_read_paths_full_capacity_pass(engine, monkeypatch, tune):
    """The pather keeps room for 20 parts / 16 edges per read in LDS and hands reads that need more to a full-capacity second
    pass; SNK_PATH_REDO_ALL sends EVERY read through it: same paths, same duplicate flags, same barcode lists."""
    c = goldens.load("adversarial")
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
    off0, ne0, edges0, info0 = res.path_reads(rows, c.read_len, quals, lens=lens, mark_dups=True, bc=bc, unitig_bcs=True)
    tune("SNK_PATH_REDO_ALL", "1")
    off, ne, edges, info = res.path_reads(rows, c.read_len, quals, lens=lens, mark_dups=True, bc=bc, unitig_bcs=True)
    assert np.array_equal(ne.astype(np.int64), c.exp_path_n) and np.array_equal(edges, c.exp_path_edges) and np.array_equal(off, c.exp_path_off)
    assert np.array_equal(info["dups"]["dup"], c.exp_dup)
    assert np.array_equal(info["unitig_bcs"][0], info0["unitig_bcs"][0]) and np.array_equal(info["unitig_bcs"][1], info0["unitig_bcs"][1])
    # and the group kernel doing the sequential tail itself (SNK_PATH_FUSED) instead of the one-thread-per-read finishing kernel
    engine.clear_option("path_redo_all")
    tune("SNK_PATH_FUSED", "1")
    off, ne, edges, info = res.path_reads(rows, c.read_len, quals, lens=lens, mark_dups=True, bc=bc, unitig_bcs=True)
    assert np.array_equal(ne.astype(np.int64), c.exp_path_n) and np.array_equal(edges, c.exp_path_edges) and np.array_equal(off, c.exp_path_off)
    assert np.array_equal(info["dups"]["dup"], c.exp_dup)


def test_unitig_barcode_lists_two_derivations_200k_parity_unpinned_rust(engine, monkeypatch, tune):
    """f4, parity unpinned (the reference side is Rust): the per-unitig barcode lists out of the pather's parts against a second,
    independent derivation ON THE DEVICE -- every k-mer of every barcoded read looked up in the f1 dictionary, which is literally
    barcodes_for_sedge (debruijn.rs:115-131) + the union of cmd_main_asm.rs:91-151 -- on 200 k synthetic reads; and the 20 000-entry
    cut (cmd_main_asm.rs:115; here: the smallest ids are kept) with the cut lowered to 3."""
    import torch
    from supernova_amd import synth
    sp = synth.synth_params(200_000, seed=0x5EED0B0C)
    rows, quals, bc = engine.synth(sp)
    res = engine.count_graph(rows, sp.read_len, quals=quals, bc=bc)
    _, _, _, fast = res.path_reads(rows, sp.read_len, quals, bc=bc, unitig_bcs=True)
    _, _, _, slow = res.path_reads(rows, sp.read_len, quals, bc=bc, unitig_bcs="exhaustive")
    assert len(fast["unitig_bcs"][1]) > 1000
    assert np.array_equal(fast["unitig_bcs"][0], slow["unitig_bcs"][0]) and np.array_equal(fast["unitig_bcs"][1], slow["unitig_bcs"][1])
    tune("SNK_UNITIG_BC_CUT", "3")
    _, _, _, cut = res.path_reads(rows, sp.read_len, quals, bc=bc, unitig_bcs=True)
    _, _, _, nocut = res.path_reads(rows, sp.read_len, quals, bc=bc, unitig_bcs=True, bcs_nocut=True)
    assert np.array_equal(nocut["unitig_bcs"][0], fast["unitig_bcs"][0]) and np.array_equal(nocut["unitig_bcs"][1], fast["unitig_bcs"][1])
    off, b = fast["unitig_bcs"]
    coff, cb = cut["unitig_bcs"]
    assert int(np.diff(off.astype(np.int64)).max()) > 3 and int(np.diff(coff.astype(np.int64)).max()) == 3
    for u in range(len(off) - 1):
        want = b[int(off[u]):int(off[u + 1])][:3]
        assert np.array_equal(cb[int(coff[u]):int(coff[u + 1])], want)


def test_read_paths_index_equals_kmer_dictionary_200k_k60(engine, monkeypatch, tune):
    """The two look-up structures of the pather -- minimiser index (places a unitig k-mer's window picks, verified against the packed
    unitigs) and k-mer dictionary (a slot per unitig k-mer) -- give the same paths, duplicate flags and barcode lists on 200 k reads with
    0.6 % errors at K=48 and on the K=60 golden graph."""
    from supernova_amd import synth
    from supernova_amd.engine import Params
    sp = synth.synth_params(200_000, seed=0x5EED0B1D, sub_ppm=6000)
    rows, quals, bc = engine.synth(sp)
    res = engine.count_graph(rows, sp.read_len, quals=quals, bc=bc)
    out = {}
    for mode in ("1", "0"):
        tune("SNK_PATH_INDEX", mode)
        off, ne, edges, info = res.path_reads(rows, sp.read_len, quals, bc=bc, mark_dups=True, unitig_bcs=True)
        out[mode] = (off, ne, edges, info["dups"]["dup"], info["unitig_bcs"][0], info["unitig_bcs"][1])
    assert int((out["1"][1] > 0).sum()) > 150_000
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a, b)
    c = goldens.load("adversarial")
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=None, lens=lens, params=Params(K=60))
    out = {}
    for mode in ("1", "0"):
        tune("SNK_PATH_INDEX", mode)
        off, ne, edges, info = res.path_reads(rows, c.read_len, quals, lens=lens)
        out[mode] = (off, ne, edges)
    assert int((out["1"][1] > 0).sum()) > 0
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("mask", ["0xFF", "0x3"])
def test_read_paths_with_colliding_fingerprints(engine, monkeypatch, mask, tune):
    """The dictionary keeps a 64-bit fingerprint per k-mer (16-byte slots) and the pather checks a match against the unitig's
    bases.  SNK_PATH_FP_MASK narrows the fingerprint to 8 / 2 bits: nearly every probe chain now holds false matches, the reads fall
    back to the verified look-up -- and the paths are still the reference's, bit for bit."""
    tune("SNK_PATH_FP_MASK", mask)
    tune("SNK_PATH_INDEX", "0")          # (the minimiser index compares bases on every look-up: it has no fingerprints)
    for name in ("adversarial", "synth_20k_err"):
        c = goldens.load(name)
        rows, quals, bc, lens = _to_dev(c)
        res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
        off, ne, edges, info = res.path_reads(rows, c.read_len, quals, lens=lens)
        assert np.array_equal(ne.astype(np.int64), c.exp_path_n) and np.array_equal(edges, c.exp_path_edges) and np.array_equal(off, c.exp_path_off)


@pytest.mark.parametrize("name", ["synth_2k_err", "adversarial", "synth_4k_dups"])
def test_unitig_barcode_lists_parity_unpinned_rust(engine, graph_stage, name):
    """The rest of f4: per-unitig barcode lists out of the pather's exact-match parts, against the plain-Python restatement of
    tada's edge -> barcode sets (every k-mer of every barcoded read looked up).  Parity unpinned: the Rust reference cannot be
    built here, the restatement follows lib/tada/src/cmd_main_asm.rs:91-151 / debruijn.rs:115-131."""
    if graph_stage == "global":
        pytest.skip("one graph stage is enough")
    c = goldens.load(name)
    rows, quals, bc, lens = _to_dev(c)
    res = engine.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, ign_bc_below=c.ign_bc_below)
    uoff, ubases = res.unitig_arrays()
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[ubases].tobytes().decode()
    us = [asc[int(uoff[i]):int(uoff[i + 1])] for i in range(res.n_unitigs)]
    off, ne, edges, info = res.path_reads(rows, c.read_len, quals, lens=lens, bc=bc, unitig_bcs=True)
    assert np.array_equal(ne.astype(np.int64), c.exp_path_n) and np.array_equal(edges, c.exp_path_edges)      # the paths are untouched by it
    boff, bcs = info["unitig_bcs"]
    exp = oracle_lib.unitig_barcodes(c.codes, c.lens, c.bc, us, K=48)
    assert len(boff) == len(us) + 1 and int(boff[-1]) == len(bcs) == sum(len(x) for x in exp)
    for u in range(len(us)):
        assert bcs[int(boff[u]):int(boff[u + 1])].tolist() == exp[u], u
    assert len(bcs) > 0


@pytest.mark.parametrize("min_bc", [3, 5])
def test_minbc_above_two_synth(engine, graph_stage, min_bc):
    """General minBC on a seeded workload with many barcodes per locus (40 barcodes over 60 k reads: every locus sees several),
    one GPU and the count kernel's hash-split path (3 buckets), against the C oracle."""
    from supernova_amd import synth
    from supernova_amd.engine import Params
    if graph_stage == "global":
        pytest.skip("the barcode rule lives in the count kernel")
    n = 60_000
    sp = synth.synth_params(n, seed=0x5EED0B00 + min_bc, pairs_per_bc=750)
    rows_h, quals_h, bc_h = synth.synth_host(sp, qstride=160)
    rows_d, quals_d, bc_d = engine.synth(sp, qstride=160)
    gl = oracle_lib.good_lens(quals_h, 150)
    o = oracle_lib.OracleResult(synth.unpack_rows(rows_h, 150), gl, bc_h, min_bc=min_bc, hbv=False)
    o2 = oracle_lib.OracleResult(synth.unpack_rows(rows_h, 150), gl, bc_h, min_bc=2, hbv=False)
    assert 0 < o.keys.shape[0] < o2.keys.shape[0]                 # the stricter rule really drops k-mers
    hist = np.bincount(np.minimum(o.counts, (1 << 24) - 1)).astype(np.int64)
    for nb in (0, 3):
        res = engine.count_graph(rows_d, 150, quals=quals_d, bc=bc_d, params=Params(K=48, min_bc=min_bc, n_buckets=nb))
        _check_against(res, o.keys[:, :3], o.counts, o.ctx, o.unitigs, gl, hist)
