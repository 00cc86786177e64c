"""GPU parity of the minimiser-sharded path: W simulated ranks (threads, one context each) on ONE GPU run the
same SPMD code as the multi-GPU bench, with tensor copies in place of the RCCL transport.  The union of the
ranks' tables and of the unitigs every rank wrote (owner-side join) must equal the reference's golden vectors."""
import threading

import numpy as np
import pytest

import goldens

pytestmark = pytest.mark.gpu


def run_world(W, case, n_buckets=0, long_minimiser=False):
    import torch
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine, SimWorld

    dev = torch.device("cuda", 0)
    world = SimWorld(W)
    n = case.rows.shape[0]
    bounds = [n * r // W for r in range(W + 1)]
    out, errs = [None] * W, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            e = Engine(0)
            lo, hi = bounds[r], bounds[r + 1]
            rows = torch.from_numpy(case.rows[lo:hi].view(np.int32).copy()).to(dev)
            quals = torch.from_numpy(np.ascontiguousarray(case.quals[lo:hi])).to(dev)
            bc = torch.from_numpy(case.bc[lo:hi].astype(np.int32)).to(dev)
            lens = torch.from_numpy(case.lens[lo:hi].astype(np.uint16).view(np.int16)).to(dev)
            sh = ShardedEngine(e, world.comm(r))
            res = sh.count_graph(rows, case.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48, n_buckets=n_buckets, long_minimiser=long_minimiser),
                                 ign_bc_below=case.ign_bc_below, read_index_base=lo)
            out[r] = dict(keys=res.keys(), counts=res.counts(), ctx=res.ctx(), spectrum=res.spectrum(),
                          n_instances=res.n_instances, n_frags=res.n_frags, n_queries=res.n_queries,
                          unitigs=res.unitigs(), ranking=getattr(res, "join_ranking", None), n_hot=int(res.raw.n_hot_buckets))
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)
            world.barrier_obj.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    return out


def all_unitigs(out):
    """Union of the unitigs the ranks wrote, in the reference's BVComp order."""
    return sorted((u for o in out for u in o["unitigs"]), key=lambda s: (-len(s), s))


def check(out, c):
    keys = np.concatenate([o["keys"] for o in out])
    counts = np.concatenate([o["counts"] for o in out])
    ctx = np.concatenate([o["ctx"] for o in out])
    order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    keys, counts, ctx = keys[order], counts[order], ctx[order]
    assert keys.shape[0] == c.exp_keys.shape[0]
    assert np.array_equal(keys[:, :3], c.exp_keys) and np.all(keys[:, 3] == 0)
    assert np.array_equal(np.minimum(counts, (1 << 24) - 1), c.exp_counts)
    assert np.array_equal(ctx, c.exp_ctx)
    nb = max(len(o["spectrum"]) for o in out)       # ranks size their spectra by their own largest count
    spec = sum(np.pad(o["spectrum"].astype(np.int64), (0, nb - len(o["spectrum"]))) for o in out)
    nz = np.nonzero(spec)[0]
    assert np.array_equal(spec[: (nz[-1] + 1 if len(nz) else 0)], c.exp_hist)
    assert all_unitigs(out) == c.exp_unitigs
    tot_inst = sum(o["n_instances"] for o in out)
    exp_inst = int(sum(int(g) - 47 for g in c.exp_goodlens if g >= 49))
    assert tot_inst == exp_inst


@pytest.mark.parametrize("W", [1, 2, 3])
@pytest.mark.parametrize("name", ["adversarial", "synth_20k_err"])
def test_sharded_matches_reference(snk, W, name):
    c = goldens.load(name)
    out = run_world(W, c)
    check(out, c)
    if W > 1:
        assert sum(o["n_queries"] for o in out) > 0      # the cross-rank prune really ran
    # the ranking is partitioned over the ranks -- also when fragment lists are circles (the plasmids of the adversarial case,
    # one of them without any splitter): every rank cuts them the same way in the replicated links and ranks again
    assert {o["ranking"] for o in out} == {"partitioned"}


@pytest.mark.parametrize("W", [1, 2, 3])
@pytest.mark.parametrize("name", ["adversarial", "synth_20k_err"])
def test_sharded_long_minimisers(snk, W, name):
    """SNK_F_LONG_MINIMISER through the N-rank step: the partition, the neighbour classification of the prune and the owner of a remote
    k-mer (the cross-rank queries) all use the 20-base minimiser; same results as the reference."""
    c = goldens.load(name)
    out = run_world(W, c, long_minimiser=True)
    check(out, c)
    if W > 1:
        assert sum(o["n_queries"] for o in out) > 0


@pytest.mark.parametrize("W", [1, 3])
def test_sharded_replicated_ranking_with_circles(snk, W, monkeypatch, tune):
    """SNK_JOIN_REPLICATED=1: every rank ranks the whole link structure itself; the circles are cut by the same sparse pass."""
    tune("SNK_JOIN_REPLICATED", "1")
    c = goldens.load("adversarial")
    out = run_world(W, c)
    check(out, c)
    assert {o["ranking"] for o in out} == {"replicated"}


@pytest.mark.parametrize("W", [1, 3])
def test_sharded_overflowing_buckets_and_the_hot_table(snk, W, monkeypatch, tune):
    """Tiny bucket capacity: most supermers travel through the overflow list, and buckets noted as hot stop counting in their cursors
    (snk_msp.hip) -- the per-bucket histogram the exchange is planned from is rebuilt from the grouped overflow list."""
    tune("SNK_MSP_CAP_PCT", "10")
    tune("SNK_MSP_HOT_FACTOR", "1")
    tune("SNK_MSP_HOT_MIN", "1")
    c = goldens.load("synth_20k_err")
    check(run_world(W, c), c)


@pytest.mark.parametrize("W", [1, 2, 3])
@pytest.mark.parametrize("name", ["adversarial", "synth_20k_err"])
def test_sharded_hot_buckets_are_repartitioned_on_their_owner(snk, W, name, monkeypatch, tune):
    """snk_hot.hip on a rank of the N-GPU job: a minimiser bucket's records arrive as one segment per source rank (+ the owner's own slots
    and overflow); a bucket far above its capacity is planned from the exchanged histograms, its records are expanded into hash classes
    when the exchange is through, and the classes are counted by a launch of their own.  Forced on the goldens by a tiny threshold."""
    tune("SNK_MSP_CAP_PCT", "20")
    tune("SNK_HOT_MIN", "8")
    tune("SNK_HOT_FACTOR", "1")
    tune("SNK_HOT_CLASS_INST", "300")
    c = goldens.load(name)
    out = run_world(W, c)
    check(out, c)
    assert sum(o["n_hot"] for o in out) > 0


@pytest.mark.parametrize("W", [1, 2, 3])
def test_sharded_streamed_slabs_equal_the_resident_step(snk, W):
    """snk_shard_stream_begin / _append / _finish: a rank's reads arrive in slabs of uneven size and are partitioned as they come; everything
    behind the partition is the resident step's.  Same table, contexts, spectrum and unitigs as the reference's (the golden), i.e. as the
    resident step gives."""
    import threading
    import torch
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine, SimWorld
    c = goldens.load("synth_20k_err")
    dev = torch.device("cuda", 0)
    world = SimWorld(W)
    n = c.rows.shape[0]
    bounds = [n * r // W for r in range(W + 1)]
    out, errs = [None] * W, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            e = Engine(0)
            lo, hi = bounds[r], bounds[r + 1]
            cuts = [lo, lo + (hi - lo) // 7, lo + (hi - lo) // 7, lo + (hi - lo) // 2, hi]        # (one empty slab among them)
            def slabs():
                for a, b in zip(cuts[:-1], cuts[1:]):
                    yield dict(rows=torch.from_numpy(c.rows[a:b].view(np.int32).copy()).to(dev), quals=torch.from_numpy(np.ascontiguousarray(c.quals[a:b])).to(dev),
                               bc=torch.from_numpy(c.bc[a:b].astype(np.int32)).to(dev), lens=torch.from_numpy(c.lens[a:b].astype(np.uint16).view(np.int16)).to(dev),
                               read_index_base=a)
            sh = ShardedEngine(e, world.comm(r))
            res = sh.count_graph_streamed(slabs(), c.read_len, total_reads=n, rank_reads_ub=hi - lo + 5, params=Params(K=48), ign_bc_below=c.ign_bc_below)
            out[r] = dict(keys=res.keys(), counts=res.counts(), ctx=res.ctx(), spectrum=res.spectrum(), n_instances=res.n_instances, n_frags=res.n_frags,
                          n_queries=res.n_queries, unitigs=res.unitigs())
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)
            world.barrier_obj.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    check(out, c)


def test_sharded_many_small_buckets(snk):
    c = goldens.load("adversarial")
    check(run_world(4, c, n_buckets=4 * 997), c)


@pytest.mark.parametrize("n_buckets", [0, 8 * 3])
def test_sharded_eight_ranks(snk, n_buckets):
    """Eight simulated ranks: every bucket is counted from eight record segments as one concatenated stream (ranged
    exchange, four count launches); with 24 buckets in all, batches span several segments and the buckets split."""
    c = goldens.load("synth_20k_err")
    out = run_world(8, c, n_buckets=n_buckets)
    check(out, c)


def test_sharded_engine_reuse_across_inputs(snk):
    """One engine + ShardedEngine per rank (pooled exchange buffers, cached arena) over inputs of changing size: every
    call is checked against the reference's golden vectors -- stale buffer contents or sizes would show."""
    import torch
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine, SimWorld
    W = 2
    world = SimWorld(W)
    dev = torch.device("cuda", 0)
    names = ["synth_20k_err", "synth_2k_err", "adversarial", "synth_6k_clean", "synth_20k_err", "synth_2k_err"]
    cases = [goldens.load(nm) for nm in names]
    outs, errs = [[None] * W for _ in names], []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            e = Engine(0)
            sh = ShardedEngine(e, world.comm(r))
            for ci, c in enumerate(cases):
                n = c.rows.shape[0]
                lo, hi = n * r // W, n * (r + 1) // W
                rows = torch.from_numpy(c.rows[lo:hi].view(np.int32).copy()).to(dev)
                quals = torch.from_numpy(np.ascontiguousarray(c.quals[lo:hi])).to(dev)
                bc = torch.from_numpy(c.bc[lo:hi].astype(np.int32)).to(dev)
                lens = torch.from_numpy(c.lens[lo:hi].astype(np.uint16).view(np.int16)).to(dev)
                res = sh.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48),
                                     ign_bc_below=c.ign_bc_below, read_index_base=lo)
                outs[ci][r] = dict(keys=res.keys(), counts=res.counts(), ctx=res.ctx(), spectrum=res.spectrum(),
                                   n_instances=res.n_instances, n_frags=res.n_frags, n_queries=res.n_queries,
                                   unitigs=res.unitigs())
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)
            world.barrier_obj.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    for ci, c in enumerate(cases):
        check(outs[ci], c)


def test_sharded_k60(snk):
    """K=60 through the sharded path (2 ranks) against the C oracle."""
    import oracle_lib
    c = goldens.load("adversarial")
    import torch
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine, SimWorld
    W = 2
    world = SimWorld(W)
    dev = torch.device("cuda", 0)
    n = c.rows.shape[0]
    bounds = [n * r // W for r in range(W + 1)]
    out, errs = [None] * W, []

    def worker(r):
        try:
            e = Engine(0)
            lo, hi = bounds[r], bounds[r + 1]
            res = ShardedEngine(e, world.comm(r)).count_graph(
                torch.from_numpy(c.rows[lo:hi].view(np.int32).copy()).to(dev), c.read_len,
                quals=torch.from_numpy(np.ascontiguousarray(c.quals[lo:hi])).to(dev),
                bc=torch.from_numpy(c.bc[lo:hi].astype(np.int32)).to(dev),
                lens=torch.from_numpy(c.lens[lo:hi].astype(np.uint16).view(np.int16)).to(dev),
                params=Params(K=60), ign_bc_below=c.ign_bc_below, read_index_base=lo)
            out[r] = dict(keys=res.keys(), counts=res.counts(), ctx=res.ctx(), unitigs=res.unitigs())
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)
            world.barrier_obj.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errs:
        raise errs[0]
    gl = oracle_lib.good_lens(c.quals, c.lens, K=60)
    o = oracle_lib.OracleResult(c.codes, gl, c.bc, K=60, ign_bc_below=c.ign_bc_below, hbv=False)
    keys = np.concatenate([x["keys"] for x in out])
    order = np.lexsort((keys[:, 3], keys[:, 2], keys[:, 1], keys[:, 0]))
    assert np.array_equal(keys[order], o.keys)
    assert np.array_equal(np.concatenate([x["counts"] for x in out])[order], o.counts)
    assert np.array_equal(np.concatenate([x["ctx"] for x in out])[order], o.ctx)
    assert all_unitigs(out) == o.unitigs


def test_rccl_world1_exchange_multi_gib(snk):
    """A real RCCL group of one rank: the exchange must deliver every byte of a multi-GiB segment.  (RCCL 2.26 moves only the
    first half of a > 1 GiB message a rank sends to itself through all_to_all_single; TorchComm copies that segment itself and
    cuts the others into pieces.)"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from supernova_amd.sharded import TorchComm
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dev = torch.device("cuda", 0)
        n = (5 << 29) + 4096 * 3            # 2.5 GiB + a ragged tail
        src = torch.empty(n, dtype=torch.uint8, device=dev)
        src.view(torch.int32).copy_(torch.arange(n // 4, dtype=torch.int32, device=dev))
        recv, rb = TorchComm(dist).all_to_all_v(src, [n])
        torch.cuda.synchronize()
        assert rb == [n] and torch.equal(recv, src)
    finally:
        dist.destroy_process_group()


def _plasmid_case(seed=5):
    """A 600 kb linear genome + six circular replicons from 150 bp to 20 kb at 30x, error-free, two barcodes per locus:
    thousands of fragments (the sparse-ruling-set ranking runs, not the small-input fallback) with circles that hold splitters
    and circles that hold none."""
    rng = np.random.default_rng(seed)
    L = 150
    reps = [(rng.integers(0, 4, 600_000, dtype=np.uint8), False)] + [(rng.integers(0, 4, n, dtype=np.uint8), True) for n in (150, 400, 1000, 3000, 8000, 20000)]
    rows = []
    for g, circular in reps:
        G = len(g)
        n = max(40, G * 30 // L)
        ext = np.concatenate([g, g[:L]]) if circular else g
        starts = rng.integers(0, G if circular else G - L + 1, n)
        idx = starts[:, None] + np.arange(L)[None, :]
        r = ext[idx]
        flip = rng.random(n) < 0.5
        r[flip] = (3 - r[flip][:, ::-1])
        rows.append(r)
    codes = np.concatenate(rows).astype(np.uint8)
    perm = rng.permutation(codes.shape[0])
    codes = codes[perm]
    if codes.shape[0] & 1:
        codes = codes[:-1]
    n = codes.shape[0]
    quals = np.full((n, L), 30, dtype=np.uint8)
    bc = rng.integers(1, 50, n).astype(np.int32)
    return codes, quals, bc, L


@pytest.mark.parametrize("W", [1, 2, 3])
def test_sharded_circles_at_scale_against_the_oracle(snk, W):
    """Circular replicons among tens of thousands of fragments: the partitioned ranking cuts them (splitter cycles by pointer
    jumping on the splitter list, splitter-free circles by walking them) and still ranks partitioned; table and unitigs equal
    the C oracle's, so does the one-GPU path (whose join uses the same cut instead of Wyllie's jumping over all states)."""
    import oracle_lib
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine, SimWorld
    codes, quals, bc, L = _plasmid_case()
    n = codes.shape[0]
    o = oracle_lib.OracleResult(codes, np.full(n, L, np.uint32), bc, hbv=False)
    assert sum(1 for u in o.unitigs if len(u) >= 95 and u[:47] == u[-47:]) >= 6
    rows = synth.pack_rows(codes)
    dev = torch.device("cuda", 0)
    world = SimWorld(W)
    bounds = [(n // 2 * r // W) * 2 for r in range(W)] + [n]
    out, errs = [None] * W, []

    def worker(r):
        try:
            e = Engine(0)
            lo, hi = bounds[r], bounds[r + 1]
            res = ShardedEngine(e, world.comm(r)).count_graph(
                torch.from_numpy(rows[lo:hi].view(np.int32).copy()).to(dev), L, quals=torch.from_numpy(quals[lo:hi].copy()).to(dev),
                bc=torch.from_numpy(bc[lo:hi].copy()).to(dev), params=Params(K=48), read_index_base=lo, total_reads=n)
            out[r] = dict(keys=res.keys(), counts=res.counts(), ctx=res.ctx(), unitigs=res.unitigs(), ranking=res.join_ranking, n_circles=res.n_circles,
                          n_frags=res.n_frags)
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)
            world.barrier_obj.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errs:
        raise errs[0]
    assert sum(x["n_frags"] for x in out) > 4096
    keys = np.concatenate([x["keys"] for x in out])
    order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    assert np.array_equal(keys[order][:, :3], o.keys[:, :3])
    assert np.array_equal(np.concatenate([x["counts"] for x in out])[order], o.counts)
    assert np.array_equal(np.concatenate([x["ctx"] for x in out])[order], o.ctx)
    assert all_unitigs(out) == o.unitigs
    assert {x["ranking"] for x in out} == {"partitioned"} and all(x["n_circles"] >= 6 for x in out)
    if W == 1:
        e = Engine(0)
        res = e.count_graph(torch.from_numpy(rows.view(np.int32).copy()).to(dev), L, quals=torch.from_numpy(quals).to(dev), bc=torch.from_numpy(bc).to(dev),
                            params=Params(K=48))
        assert res.unitigs() == o.unitigs and res.n_circles >= 6
        e.close()


@pytest.mark.parametrize("n", [1_200_000, 3_400_000])
def test_sharded_bucket_size_follows_the_data(snk, n):
    """Error-rich reads: the second step of a sharded job sizes its buckets from the job-wide ratio of distinct k-mers per instance
    the first step exchanged (the same decision on every rank); a first step with enough buckets to look at (the larger case)
    sees its first buckets overflow, the ranks agree, and it partitions and exchanges a second time itself.  Either way the same
    table and unitigs as the one-GPU path."""
    import math
    import threading
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine, SimWorld
    W = 2
    sp = synth.synth_params(n, seed=0x5EED0E78, sub_ppm=15000)
    lam, term, cum = 150 * 15000 / 1e6, math.exp(-150 * 15000 / 1e6), 0.0
    for j in range(4):
        cum += term
        sp.err_cdf[j] = min(0xFFFFFFFF, int(cum * 4294967296.0))
        term *= lam / (j + 1)
    e0 = Engine(0)
    rows, quals, bc = e0.synth(sp)
    ref = e0.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
    ref_keys, ref_counts, ref_unitigs = ref.keys(), ref.counts(), sorted(ref.unitigs())
    world = SimWorld(W)
    outs, errs = [[None] * W for _ in range(2)], []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            e = Engine(0)
            sh = ShardedEngine(e, world.comm(r))
            lo, hi = n * r // W, n * (r + 1) // W
            for step in range(2):
                res = sh.count_graph(rows[lo:hi].contiguous(), 150, quals=quals[lo:hi].contiguous(), bc=bc[lo:hi].contiguous(), params=Params(K=48),
                                     read_index_base=lo, total_reads=n)
                outs[step][r] = dict(keys=res.keys(), counts=res.counts(), unitigs=res.unitigs(), nb=int(res.raw.n_buckets_total), rep=int(res.raw.repartitioned), lim=e.last_count_limit())
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)
            world.barrier_obj.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    for step in range(2):
        keys = np.concatenate([o["keys"] for o in outs[step]])
        counts = np.concatenate([o["counts"] for o in outs[step]])
        order = np.lexsort((keys[:, 3], keys[:, 2], keys[:, 1], keys[:, 0]))
        assert np.array_equal(keys[order], ref_keys) and np.array_equal(counts[order], ref_counts)
        assert sorted(u for o in outs[step] for u in o["unitigs"]) == ref_unitigs
        assert outs[step][0]["nb"] == outs[step][1]["nb"]
    # tables that run this full (0.4 distinct k-mers per instance): the bit filter in front of a 1024-slot table (960 usable), booked slots, on every
    # rank -- the same turn the one-GPU path takes, from the job-wide ratio
    assert all(o["lim"] == 960 for o in outs[1])
    if n < 2_000_000:
        assert outs[0][0]["lim"] == 1216
        assert outs[0][0]["rep"] == 0 and outs[1][0]["nb"] > 1.1 * outs[0][0]["nb"]
    else:
        assert outs[0][0]["rep"] == 1 and outs[0][1]["rep"] == 1 and outs[1][0]["rep"] == 0
        assert outs[0][0]["nb"] * W * 5000 > 1.5 * n * 103 and abs(outs[1][0]["nb"] - outs[0][0]["nb"]) < 0.2 * outs[0][0]["nb"]
    e0.close()
