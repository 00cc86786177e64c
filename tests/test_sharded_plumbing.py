"""CPU tests of the sharded host logic: bucket planning, ownership, segment offsets, the circle rule, and the
torch.distributed exchanges on the gloo backend with world_size 2 (the N>1 plumbing without GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import goldens
import oracle_lib


def test_plan_and_ownership():
    from supernova_amd.sharded import owner_record_counts, plan_buckets, segment_offsets
    nb = plan_buckets(10_300_000_000, 8, 48)
    assert nb % 8 == 0 and nb >= 8
    hist = torch.tensor([3, 0, 2, 5, 1, 1, 0, 4], dtype=torch.int32)
    off = torch.zeros(9, dtype=torch.int64)
    off[1:] = torch.cumsum(hist.to(torch.int64), 0)
    assert owner_record_counts(off, 2) == [10, 6]
    assert owner_record_counts(off, 4) == [3, 7, 2, 4]
    hr = torch.tensor([[3, 0, 2, 5], [1, 1, 0, 0]], dtype=torch.int32)       # two sources, my 4 buckets
    seg = segment_offsets(hr, [10, 2])
    assert seg.tolist() == [[0, 3, 3, 5, 10], [10, 11, 12, 12, 12]]


def test_canonicalize_circle_matches_reference_rule():
    """Cut the reference's circular unitigs (golden 'adversarial' holds two plasmids) at every rotation and strand:
    the host rule must give back the reference's sequence."""
    from supernova_amd.sharded import canonicalize_circle
    c = goldens.load("adversarial")
    K = 48
    lut = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
    found = 0
    for u in c.exp_unitigs:
        if len(u) >= 2 * K - 1 and u[: K - 1] == u[-(K - 1):] and len(u) - (K - 1) in (700, 61):
            codes = np.array([lut[ord(ch)] for ch in u], dtype=np.uint8)
            n = len(codes) - (K - 1)
            ring = codes[:n]
            for strand in (0, 1):
                r = ring if strand == 0 else (3 - ring[::-1]).astype(np.uint8)
                for rot in range(0, n, max(1, n // 13)):
                    rr = np.concatenate([r[rot:], r[:rot]])
                    cut = np.concatenate([rr, rr[: K - 1]])
                    assert np.array_equal(canonicalize_circle(cut, K), codes)
            found += 1
    assert found == 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from supernova_amd.sharded import TorchComm, owner_record_counts, segment_offsets
    comm = TorchComm(dist)
    ok = True
    # a bucketed record exchange: NB_total buckets, records are (bucket, src, serial) triples of int32 (12 B)
    rng = np.random.default_rng(100 + rank)
    NB_total = 8
    NBl = NB_total // world
    hist = rng.integers(0, 5, NB_total).astype(np.int32)
    recs = []
    for b in range(NB_total):
        for s in range(hist[b]):
            recs.append((b, rank, s))
    recs = np.array(recs, dtype=np.int32).reshape(-1, 3)
    off = torch.zeros(NB_total + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(torch.from_numpy(hist).to(torch.int64), 0)
    counts = owner_record_counts(off, world)
    hist_recv = comm.all_to_all_equal(torch.from_numpy(hist).view(world, NBl))
    send = torch.from_numpy(recs.copy()).view(torch.uint8).view(-1)
    recv, recv_bytes = comm.all_to_all_v(send, [c * 12 for c in counts])
    seg = segment_offsets(hist_recv, [b // 12 for b in recv_bytes])
    got = recv.view(torch.int32).view(-1, 3).numpy()
    for s in range(world):
        for lb in range(NBl):
            a, b = int(seg[s, lb]), int(seg[s, lb + 1])
            blk = got[a:b]
            ok &= bool(np.all(blk[:, 0] == rank * NBl + lb)) and bool(np.all(blk[:, 1] == s))
            ok &= list(blk[:, 2]) == list(range(b - a))
    ok &= comm.allreduce_sum_int(rank + 1, torch.device("cpu")) == world * (world + 1) // 2
    ok &= comm.all_gather_int(10 + rank, torch.device("cpu")) == [10 + r for r in range(world)]
    # segments far larger than one transport message: every segment travels in many ragged pieces
    TorchComm.CHUNK_BYTES = 1000
    big = torch.arange(0, 70_001 + 13 * rank, dtype=torch.int64) * (rank + 1)
    cnt = [(big.numel() // 3) * 8, (big.numel() - big.numel() // 3) * 8]
    back, rb = comm.all_to_all_v(big.view(torch.uint8), cnt)
    for src in range(world):
        n_src = 70_001 + 13 * src
        full = torch.arange(0, n_src, dtype=torch.int64) * (src + 1)
        lo = 0 if rank == 0 else n_src // 3
        ln = n_src // 3 if rank == 0 else n_src - n_src // 3
        a = sum(rb[:src]) // 8
        ok &= rb[src] == ln * 8 and bool(torch.equal(back.view(torch.int64)[a:a + ln], full[lo:lo + ln]))
    # the ranged record exchange (counting overlaps the transfer): R bucket ranges, pieces of 1000 bytes, ranges waited
    # for one by one in order; every rank checks what it received against what the sources hold
    R, NBl2 = 3, 6
    def src_hist(s):          # records per (dest, local bucket) of source s, 32-byte records
        return np.random.default_rng(500 + s).integers(0, 40, (world, NBl2)).astype(np.int64)
    def src_send(s):          # dest-major, bucket-ascending: record k of (s -> p, bucket b) carries (s, p, b, k)
        h = src_hist(s)
        rows = [(s, p, b, k) for p in range(world) for b in range(NBl2) for k in range(h[p, b])]
        a = np.zeros((len(rows), 4), dtype=np.int64)
        a[:] = np.array(rows, dtype=np.int64).reshape(-1, 4) if rows else a
        return a
    h_me = src_hist(rank)
    send2 = torch.from_numpy(src_send(rank).copy()).view(torch.uint8).view(-1)
    off = np.zeros(world * NBl2 + 1, dtype=np.int64)
    off[1:] = np.cumsum(h_me.reshape(-1))
    bounds = [NBl2 * r // R for r in range(R + 1)]
    soff = [[int(off[p * NBl2 + b]) * 32 for b in bounds] for p in range(world)]
    hr = np.stack([src_hist(s)[rank] for s in range(world)])            # what I receive: [source, local bucket]
    seg2 = segment_offsets(torch.from_numpy(hr), [int(x) for x in hr.sum(axis=1)])
    roff = [[int(seg2[s, b]) * 32 for b in bounds] for s in range(world)]
    recv2 = torch.zeros(int(hr.sum()) * 32, dtype=torch.uint8)
    works = comm.exchange_ranged(send2, recv2, soff, roff, R)
    ok &= len(works) == R
    g2 = recv2.view(torch.int64).view(-1, 4).numpy()
    for r in range(R):
        for wk in works[r]:
            wk.wait()
        for s in range(world):
            for b in range(bounds[r], bounds[r + 1]):
                blk = g2[int(seg2[s, b]):int(seg2[s, b + 1])]
                ok &= blk.shape[0] == hr[s, b] and bool(np.all(blk[:, 0] == s)) and bool(np.all(blk[:, 1] == rank)) and bool(np.all(blk[:, 2] == b))
                ok &= list(blk[:, 3]) == list(range(blk.shape[0]))
    # ---- owner-side join plumbing: the links of every rank are all-gathered (ragged sizes, many pieces) ...
    from supernova_amd.sharded import recv_segments, route_offsets
    sizes = [8 * (3000 + 977 * r) for r in range(world)]
    mine = (torch.arange(sizes[rank] // 8, dtype=torch.int64) * 7 + rank).view(torch.uint8)
    allg = comm.all_gather_v(mine, sizes)
    a = 0
    for r in range(world):
        ok &= bool(torch.equal(allg[a:a + sizes[r]].view(torch.int64), torch.arange(sizes[r] // 8, dtype=torch.int64) * 7 + r))
        a += sizes[r]
    # ... and every fragment travels to the owner of its unitig: 32-byte headers and the bases in two all-to-alls, a header's
    # base offset relative to its source's (16-byte aligned) segment -- the arithmetic of ShardedEngine, modelled with numpy
    rs = np.random.default_rng(900 + rank)
    nf = 500 + 37 * rank
    owner = rs.integers(0, world, nf)
    flen = rs.integers(48, 130, nf)
    fto = [int((owner == p).sum()) for p in range(world)]
    bto = [int(flen[owner == p].sum()) for p in range(world)]
    hoff, boff, bpad = route_offsets(fto, bto)
    hdr = np.zeros((hoff[-1], 4), dtype=np.int64)
    sb = np.zeros(boff[-1], dtype=np.uint8)
    hc, bcur = list(hoff[:world]), list(boff[:world])
    for f in range(nf):
        p = int(owner[f])
        hdr[hc[p]] = (rank, f, flen[f], bcur[p] - boff[p])
        sb[bcur[p]:bcur[p] + flen[f]] = (np.arange(flen[f]) * 3 + f + 11 * rank) & 0xFF
        hc[p] += 1
        bcur[p] += int(flen[f])
    hin, hb = comm.all_to_all_v(torch.from_numpy(hdr).view(torch.uint8).view(-1), [c * 32 for c in fto])
    bin_, bb = comm.all_to_all_v(torch.from_numpy(sb), bpad)
    hseg, bseg = recv_segments(hb, bb)
    H = hin.view(torch.int64).view(-1, 4).numpy()
    B = bin_.numpy()
    for src in range(world):
        src_rng = np.random.default_rng(900 + src)
        nfs = 500 + 37 * src
        o2 = src_rng.integers(0, world, nfs)
        ok &= hseg[src + 1] - hseg[src] == int((o2 == rank).sum())
        for i in range(hseg[src], hseg[src + 1]):
            s_, f_, ln, bo = (int(x) for x in H[i])
            ok &= s_ == src and o2[f_] == rank
            got_b = B[bseg[src] + bo: bseg[src] + bo + ln]
            ok &= bool(np.array_equal(got_b, ((np.arange(ln) * 3 + f_ + 11 * src) & 0xFF).astype(np.uint8)))
    q.put((rank, ok, int(got.shape[0])))
    dist.destroy_process_group()


def test_gloo_world2_exchange():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for (_r, ok, _n) in res), res


def _cb_worker(rank, world, port, q):
    """The library's own exchange planning (snk_comm_selftest: histograms, ranged record exchange with the step's piece planner,
    region-routed queries and answers, ragged all-gather) over gloo: the C++ code asks for its exchanges through callbacks."""
    import ctypes as C
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from supernova_amd import lib as _lib
    lib = _lib.load()

    def view(ptr, nbytes):
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(int(nbytes), 1),)))[:int(nbytes)]

    calls = {"a2a": 0, "gather": 0}

    def a2a(_user, send, sbeg, scnt, recv, rbeg, rcnt, W):
        try:
            calls["a2a"] += 1
            s_hi = max([sbeg[p] + scnt[p] for p in range(W)] + [0])
            r_hi = max([rbeg[p] + rcnt[p] for p in range(W)] + [0])
            s_t, r_t = view(send, s_hi), view(recv, r_hi)
            if scnt[rank]:
                r_t[rbeg[rank]:rbeg[rank] + rcnt[rank]].copy_(s_t[sbeg[rank]:sbeg[rank] + scnt[rank]])
            ops = []
            for p in range(W):
                if p == rank:
                    continue
                if scnt[p]:
                    ops.append(dist.P2POp(dist.isend, s_t[sbeg[p]:sbeg[p] + scnt[p]].clone(), p))
                if rcnt[p]:
                    ops.append(dist.P2POp(dist.irecv, r_t[rbeg[p]:rbeg[p] + rcnt[p]], p))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return 0
        except BaseException:      # a ctypes callback must not raise
            import traceback
            traceback.print_exc()
            return 1

    def gather(_user, mine, k, allp, W):
        try:
            calls["gather"] += 1
            m = view(mine, 8 * k).view(torch.int64).clone()
            outs = [torch.empty_like(m) for _ in range(W)]
            dist.all_gather(outs, m)
            dst = view(allp, 8 * k * W).view(torch.int64)
            for p in range(W):
                dst[p * k:(p + 1) * k].copy_(outs[p])
            return 0
        except BaseException:
            import traceback
            traceback.print_exc()
            return 1

    cb_a, cb_g = _lib.COMM_A2A(a2a), _lib.COMM_GATHER(gather)
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    ok = lib.snk_comm_create_callbacks(rank, world, cb_a, cb_g, None, C.byref(h), err, 512) == 0
    ok = ok and lib.snk_comm_kind(h) == b"callbacks" and lib.snk_comm_world(h) == world and lib.snk_comm_rank(h) == rank
    detail = ""
    for seed, nbl, R in ((1, 12, 4), (2, 7, 3), (3, 1, 1), (4, 64, 8)):
        rc = lib.snk_comm_selftest(h, seed, nbl, R, err, 512)
        if rc:
            ok = False
            detail = f"seed {seed}: check {rc}: {err.value.decode(errors='replace')}"
            break
    lib.snk_comm_destroy(h)
    q.put((rank, ok and calls["a2a"] > 10 and calls["gather"] >= 4, detail))
    dist.destroy_process_group()


def test_gloo_world2_library_exchange_planning():
    """world_size 2 on CPU: libsnk's exchange patterns through a callbacks communicator carried by gloo."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_cb_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for (_r, ok, _d) in res), res


def test_group_partition_balanced_and_contiguous():
    """C5 (per-barcode graphs) shards without a collective: contiguous group ranges balanced by read count."""
    from supernova_amd.grouped import partition_groups, read_slab
    rng = np.random.default_rng(3)
    sizes = rng.integers(1, 2000, 5000)
    for world in (1, 2, 3, 8):
        b = partition_groups(sizes, world)
        assert b[0] == 0 and b[-1] == len(sizes) and np.all(np.diff(b) >= 0)
        loads = [sizes[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(loads) - min(loads) <= 2 * sizes.max()
        bci = np.concatenate([[0], np.cumsum(sizes)])
        slabs = [read_slab(bci, b, r) for r in range(world)]
        assert slabs[0][0] == 0 and slabs[-1][1] == sizes.sum()
        assert all(slabs[r][1] == slabs[r + 1][0] for r in range(world - 1))
    assert list(partition_groups(np.array([5]), 4)) == [0, 0, 0, 0, 1] or partition_groups(np.array([5]), 4)[-1] == 1
