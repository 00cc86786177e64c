"""Minimiser-sharded multi-GPU path: the Python face of `snk_shard_step` (include/snk.h).

The step itself -- partition, histogram + record exchange (in bucket ranges, overlapped with the count), count,
cross-rank prune, fragments, fragment links, owner-side join -- runs in C++ behind the C ABI
(supernova_amd/csrc/snk_shard_step.hip) over a communicator (supernova_amd/csrc/snk_comm.hip):
  * RCCL over xGMI, one process per GPU: `ShardedEngine(engine, torch.distributed)` creates the communicator from a
    unique id that rank 0 makes and torch.distributed broadcasts (the only thing torch.distributed does here);
  * in-process ranks on ONE device (`local_world(W)`): the same SPMD code with device copies as the wire -- how N > 1 is
    tested on one GPU.
Reference counterpart: the shardio exchange files + SHARD_ASM chunks + MAIN_ASM_SN of lib/tada
(rust-shardio/src/shard.rs:184-211,488-493; cmd_shard_asm.rs:37-94; cmd_main_asm.rs:25-89).  The C++ host program that
runs the same step without Python is supernova_amd/csrc/host/snk_asm_sn.cc.

`TorchComm` and the planning helpers below are the host-side arithmetic of the exchanges in plain torch; the CPU tests
drive them over gloo with world_size 2 (tests/test_sharded_plumbing.py), and `TorchComm` is the transport behind
`snk_comm_create_callbacks` there.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np
import torch

from . import lib as _lib
from .engine import Engine, Params


# ------------------------------------------------------------------------------------------------ communicators
class TorchComm:
    """Exchanges over torch.distributed (RCCL on the GPU box, gloo in the CPU tests): grouped point-to-point sends."""

    def __init__(self, dist):
        self.dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def allreduce_sum_int(self, v: int, device) -> int:
        t = torch.tensor([v], dtype=torch.int64, device=device)
        self.dist.all_reduce(t)
        return int(t.item())

    def all_gather_int(self, v: int, device) -> list[int]:
        t = torch.tensor([v], dtype=torch.int64, device=device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [int(x.item()) for x in out]

    # Largest single message handed to the transport.  RCCL 2.26 (ROCm 7.0) moves only the first half of a message to
    # the rank itself once it exceeds 1 GiB (`all_to_all_single`, world 1, any dtype -- tools/dbg_a2a.py); the sharded
    # exchange has multi-GB segments, so every segment goes out in pieces, and the piece a rank owes itself is a plain copy.
    CHUNK_BYTES = 256 << 20

    def _a2a(self, out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits):
        """Variable all-to-all of 1-D tensors: in_splits[p] elements of `inp` go to rank p, out_splits[p] arrive from it."""
        W, me = self.world, self.rank
        ch = max(1, self.CHUNK_BYTES // inp.element_size())
        in_off = [0] * (W + 1)
        out_off = [0] * (W + 1)
        for p in range(W):
            in_off[p + 1] = in_off[p] + int(in_splits[p])
            out_off[p + 1] = out_off[p] + int(out_splits[p])
        if in_splits[me]:
            out[out_off[me]:out_off[me + 1]].copy_(inp[in_off[me]:in_off[me + 1]])
        # round j carries the j-th piece of every segment: one send and one receive per peer and group, the
        # pattern of an all-to-all (both sides derive the same piece count from the exchanged sizes)
        n_rounds = max([0] + [-(-int(x) // ch) for p in range(W) if p != me for x in (in_splits[p], out_splits[p])])
        for j in range(n_rounds):
            ops = []
            for p in range(W):
                if p == me:
                    continue
                c0 = j * ch
                if c0 < int(in_splits[p]):
                    ops.append(self.dist.P2POp(self.dist.isend, inp[in_off[p] + c0:in_off[p] + min(c0 + ch, int(in_splits[p]))], p))
                if c0 < int(out_splits[p]):
                    ops.append(self.dist.P2POp(self.dist.irecv, out[out_off[p] + c0:out_off[p] + min(c0 + ch, int(out_splits[p]))], p))
            for r in self.dist.batch_isend_irecv(ops):
                r.wait()

    def all_to_all_v(self, send: torch.Tensor, send_counts: list[int], alloc=None):
        """send: 1-D uint8; send_counts[p] bytes go to rank p.  Returns (recv uint8, recv_counts).
        alloc(nbytes) -> uint8 tensor supplies the receive buffer (default: torch.empty)."""
        dev = send.device
        sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        rc = torch.empty_like(sc)
        self._a2a(rc, sc, [1] * self.world, [1] * self.world)
        recv_counts = [int(x) for x in rc.tolist()]
        recv = alloc(sum(recv_counts)) if alloc else torch.empty(sum(recv_counts), dtype=torch.uint8, device=dev)
        # widest element type that divides every segment: multi-GB segments stay far below 2^31 elements
        for width, dt in ((8, torch.int64), (4, torch.int32), (1, torch.uint8)):
            if all(c % width == 0 for c in recv_counts) and all(c % width == 0 for c in send_counts) \
                    and send.data_ptr() % width == 0 and recv.data_ptr() % width == 0:
                break
        self._a2a(recv.view(dt), send.view(dt), [c // width for c in recv_counts], [c // width for c in send_counts])
        return recv, recv_counts

    def exchange_ranged(self, send: torch.Tensor, recv: torch.Tensor, soff, roff, n_ranges: int):
        """The record exchange cut into bucket ranges: bytes soff[p][r]..soff[p][r+1] of `send` go to rank p and land at
        roff[me-as-source...] on its side; here bytes roff[p][r]..roff[p][r+1] of `recv` arrive from rank p.  Everything is
        issued at once, range by range (so the transport delivers range 0 first); returns one list of work handles per
        range -- waiting on them makes the current stream wait for that range only (the count kernel of range r runs
        while range r+1 is still on the wire).  The rank's own share is a plain copy."""
        W, me = self.world, self.rank
        s64, r64 = send.view(torch.int64), recv.view(torch.int64)
        ch = max(1, self.CHUNK_BYTES // 8)
        if soff[me][n_ranges] > soff[me][0]:
            r64[roff[me][0] // 8: roff[me][n_ranges] // 8].copy_(s64[soff[me][0] // 8: soff[me][n_ranges] // 8])
        works = []
        for r in range(n_ranges):
            wr = []
            cnt_s = [(soff[p][r + 1] - soff[p][r]) // 8 for p in range(W)]
            cnt_r = [(roff[p][r + 1] - roff[p][r]) // 8 for p in range(W)]
            n_rounds = max([0] + [-(-x // ch) for p in range(W) if p != me for x in (cnt_s[p], cnt_r[p])])
            for j in range(n_rounds):
                ops = []
                for p in range(W):
                    if p == me:
                        continue
                    c0 = j * ch
                    if c0 < cnt_s[p]:
                        a = soff[p][r] // 8 + c0
                        ops.append(self.dist.P2POp(self.dist.isend, s64[a: a + min(ch, cnt_s[p] - c0)], p))
                    if c0 < cnt_r[p]:
                        a = roff[p][r] // 8 + c0
                        ops.append(self.dist.P2POp(self.dist.irecv, r64[a: a + min(ch, cnt_r[p] - c0)], p))
                wr += list(self.dist.batch_isend_irecv(ops))
            works.append(wr)
        return works

    def all_gather_v(self, send: torch.Tensor, counts: list[int], alloc=None) -> torch.Tensor:
        """send: 1-D uint8 of counts[me] bytes; returns the concatenation of every rank's buffer in rank order (counts = bytes
        per rank, known to everybody).  Built from the same grouped point-to-point pieces as the all-to-all."""
        W, me = self.world, self.rank
        total = sum(counts)
        out = alloc(total) if alloc else torch.empty(total, dtype=torch.uint8, device=send.device)
        off = [0] * (W + 1)
        for p in range(W):
            off[p + 1] = off[p] + int(counts[p])
        if counts[me]:
            out[off[me]:off[me + 1]].copy_(send[:counts[me]])
        for width, dt in ((8, torch.int64), (4, torch.int32), (1, torch.uint8)):
            if all(c % width == 0 for c in counts) and send.data_ptr() % width == 0 and out.data_ptr() % width == 0:
                break
        s_v, o_v = send[:counts[me]].view(dt), out.view(dt)
        ch = max(1, self.CHUNK_BYTES // width)
        n_me = counts[me] // width
        n_rounds = max([0] + [-(-(int(c) // width) // ch) for c in counts])
        for j in range(n_rounds):
            ops = []
            c0 = j * ch
            for p in range(W):
                if p == me:
                    continue
                if c0 < n_me:
                    ops.append(self.dist.P2POp(self.dist.isend, s_v[c0:min(c0 + ch, n_me)], p))
                n_p = counts[p] // width
                if c0 < n_p:
                    ops.append(self.dist.P2POp(self.dist.irecv, o_v[off[p] // width + c0: off[p] // width + min(c0 + ch, n_p)], p))
            if ops:
                for r in self.dist.batch_isend_irecv(ops):
                    r.wait()
        return out

    def all_to_all_equal(self, send: torch.Tensor) -> torch.Tensor:
        """send: [world, m] -> recv [world, m]; row p goes to rank p."""
        recv = torch.empty_like(send)
        m = send.shape[1]
        self._a2a(recv.view(-1), send.contiguous().view(-1), [m] * self.world, [m] * self.world)
        return recv

    def barrier(self):
        self.dist.barrier()


# ------------------------------------------------------------------------------------------------ planning helpers
def plan_buckets(total_inst_upper: int, world: int, K: int, target_inst: int = 0) -> int:
    target = int(target_inst) or (5000 if K == 48 else 3500)
    nb = max(1, -(-total_inst_upper // target))
    nb = min(nb, 1 << 26)
    return -(-nb // world) * world


def owner_record_counts(offsets: torch.Tensor, world: int) -> list[int]:
    """offsets: int64[NB_total+1] exclusive scan of the bucket histogram -> records per owner rank."""
    nbl = (offsets.numel() - 1) // world
    bnd = offsets[::nbl]
    return [int(x) for x in (bnd[1:] - bnd[:-1]).tolist()]


def segment_offsets(hist_recv: torch.Tensor, recv_record_counts: list[int]) -> torch.Tensor:
    """hist_recv [world, NBl] (row s = source s's supermers per local bucket) -> int64 [world, NBl+1] absolute offsets."""
    w, nbl = hist_recv.shape
    seg = torch.zeros((w, nbl + 1), dtype=torch.int64, device=hist_recv.device)
    seg[:, 1:] = torch.cumsum(hist_recv.to(torch.int64), dim=1)
    base = torch.zeros(w, dtype=torch.int64, device=hist_recv.device)
    if w > 1:
        base[1:] = torch.cumsum(torch.tensor(recv_record_counts[:-1], dtype=torch.int64, device=hist_recv.device), 0)
    return (seg + base[:, None]).contiguous()


def route_offsets(frags_to: list[int], bases_to: list[int]):
    """Owner-side join, sender: fragments / base bytes owed to every owner -> (first header per owner [W+1], first base byte per
    owner [W+1], padded base bytes per owner): every owner's bases start 16-byte aligned in the send buffer."""
    bpad = [-(-int(b) // 16) * 16 for b in bases_to]
    hoff, boff = [0], [0]
    for f, b in zip(frags_to, bpad):
        hoff.append(hoff[-1] + int(f))
        boff.append(boff[-1] + b)
    return hoff, boff, bpad


def recv_segments(hdr_bytes: list[int], base_bytes: list[int]):
    """Owner-side join, receiver: bytes that arrived from every source -> (first header per source [W+1], first base byte per
    source [W+1]); a header's base offset is relative to its source's segment."""
    hseg, bseg = [0], [0]
    for h, b in zip(hdr_bytes, base_bytes):
        hseg.append(hseg[-1] + int(h) // 32)
        bseg.append(bseg[-1] + int(b))
    return hseg, bseg


COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def canonicalize_circle(codes: np.ndarray, K: int) -> np.ndarray:
    """A closed circle sequence (n+K-1 bases, last K-1 == first K-1) cut at an arbitrary k-mer -> the reference's
    form: rotated to start at its minimum canonical k-mer in forward orientation (canonicalizeCircle,
    BuildReadQGraph48.cc:375-397), then bvec-canonical (addEdge :481-485).  numpy statement of what the join's
    jcircle_kernel does on the device; used by the tests only."""
    n = len(codes) - (K - 1)
    ring = codes[:n]
    best = None
    for strand in (0, 1):
        seq = ring if strand == 0 else COMP[ring[::-1]]
        ext = np.concatenate([seq, seq[:K - 1]])
        for i in range(n):
            km = bytes(ext[i:i + K])
            if best is None or km < best[0]:
                best = (km, strand, i)
    _, strand, i = best
    seq = ring if strand == 0 else COMP[ring[::-1]]
    rot = np.concatenate([seq[i:], seq[:i]])
    out = np.concatenate([rot, rot[:K - 1]])
    L = len(out)
    if L & 1:
        rev = bool(out[L // 2] & 2)
    else:
        rc = COMP[out[::-1]]
        rev = bytes(rc) < bytes(out)
    return COMP[out[::-1]].copy() if rev else out


# ------------------------------------------------------------------------------------------------ the SPMD step
JOIN_NAMES = ("links", "link_structure", "rank+place", "route", "emit")
PHASE_NAMES = ("partition", "compact", "exchange", "count", "prune", "fragments", "join", "total")
EXCH_NAMES = ("records", "prune_queries", "link_queries", "link_structure", "splitters", "ranks", "fragments", "total")


class ShardedResult:
    """One rank's share of a sharded step (device pointers owned by the engine's context, valid until its next call)."""

    def __init__(self, eng: Engine, K: int, raw: "_lib.SnkShardResult"):
        self._e = eng
        self.K = K
        self.raw = raw
        for f in ("n_reads", "n_instances", "n_supermers", "n_kmers", "n_unitigs", "n_frags", "n_frags_total", "n_queries",
                  "n_link_queries", "host_syncs", "buckets_split", "max_slots_used", "n_circles"):
            setattr(self, f, int(getattr(raw, f)))
        self.n_buckets = int(raw.n_buckets_total)
        self.join_ranking = "partitioned" if raw.ranking else "replicated"
        self.phase_ms = {PHASE_NAMES[i]: float(raw.phase_ms[i]) for i in range(8)}
        self.join_ms = {JOIN_NAMES[i]: float(raw.join_ms[i]) for i in range(5)}
        self.kernel_ms = {"count": float(raw.count_kernel_ms)}
        self.exchange_bytes = {EXCH_NAMES[i]: int(raw.exchanged_bytes[i]) for i in range(8)}
        # the most this rank sent to ONE other rank per exchange (xGMI is point to point: an exchange takes as long as its fullest pair)
        self.pair_max_bytes = {EXCH_NAMES[i]: int(raw.pair_max_bytes[i]) for i in range(7)}

    def _dl(self, ptr, nbytes, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        if nbytes:
            self._e._download(ptr, out.ctypes.data, nbytes)
        return out

    def keys(self):
        lohi = self._dl(self.raw.keys, self.n_kmers * 16, np.uint64, (self.n_kmers, 2))
        lo, hi = lohi[:, 0], lohi[:, 1]
        w = np.empty((self.n_kmers, 4), dtype=np.uint32)
        w[:, 0] = hi >> np.uint64(32); w[:, 1] = hi & np.uint64(0xFFFFFFFF)
        w[:, 2] = lo >> np.uint64(32); w[:, 3] = lo & np.uint64(0xFFFFFFFF)
        return w

    def counts(self):
        return self._dl(self.raw.counts, self.n_kmers * 4, np.uint32, (self.n_kmers,))

    def ctx(self):
        return self._dl(self.raw.ctx, self.n_kmers, np.uint8, (self.n_kmers,))

    def spectrum(self):
        nb = int(self.raw.spectrum_bins)
        return self._dl(self.raw.spectrum, nb * 8, np.uint64, (nb,))

    def unitig_arrays(self):
        n, tb = self.n_unitigs, int(self.raw.unitig_total_bases)
        return (self._dl(self.raw.unitig_off, (n + 1) * 8, np.uint64, (n + 1,)), self._dl(self.raw.unitig_bases, tb, np.uint8, (tb,)))

    def unitigs(self) -> list[str]:
        """Canonical unitigs this rank wrote (the ones whose head fragment it owns), sorted by BVComp; the union over the ranks
        is the data set's unitig set."""
        off, bases = self.unitig_arrays()
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        out = [lut[bases[int(off[i]):int(off[i + 1])]].tobytes().decode() for i in range(self.n_unitigs)]
        out.sort(key=lambda s: (-len(s), s))
        return out


def local_world(world: int) -> list[int]:
    """`world` in-process ranks on one device: handles for ShardedEngine(engine, handle), one per rank / host thread."""
    lib = _lib.load()
    arr = (C.c_void_p * world)()
    err = C.create_string_buffer(512)
    rc = lib.snk_comm_create_local(world, arr, err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    return [int(arr[r]) for r in range(world)]


class SimWorld:
    """W in-process ranks on one GPU (tests, tools): `comm(r)` is rank r's communicator handle for ShardedEngine; every rank
    runs on its own host thread with its own Engine.  barrier_obj.abort() releases the ranks that wait in an exchange after
    another one failed outside the step."""

    class _Abort:
        def __init__(self, w):
            self.w = w

        def abort(self):
            lib = _lib.load()
            for h in self.w.handles:
                lib.snk_comm_abort(h)

    def __init__(self, world: int, serial: bool = False):
        self.world = world
        self.handles = local_world(world)
        self.barrier_obj = SimWorld._Abort(self)

    def comm(self, rank: int) -> int:
        return self.handles[rank]


def rccl_comm(engine: Engine, dist) -> int:
    """An RCCL communicator for libsnk on the engine's device: rank 0 makes the 128-byte unique id, torch.distributed carries
    it to the other ranks (a broadcast of 128 bytes -- everything else of the step runs behind the C ABI)."""
    lib = engine.lib
    err = C.create_string_buffer(512)
    rank, world = dist.get_rank(), dist.get_world_size()
    from pathlib import Path
    cand = Path(torch.__file__).parent / "lib" / "librccl.so"      # the RCCL of this process: the one torch ships
    if cand.exists():
        lib.snk_comm_set_rccl_path(str(cand).encode())
    idb = (C.c_uint8 * 128)()
    if rank == 0:
        rc = lib.snk_comm_unique_id(idb, err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    if world > 1:
        dev = torch.device("cuda", engine.device) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor(list(idb), dtype=torch.uint8, device=dev)
        dist.broadcast(t, 0)
        for i, v in enumerate(t.cpu().tolist()):
            idb[i] = v
    h = C.c_void_p()
    rc = lib.snk_comm_create_rccl(engine._ctx, idb, rank, world, C.byref(h), err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    return int(h.value)


class _DevBytes:
    """A device array of the library seen by torch without a copy (__cuda_array_interface__ over the raw pointer)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def gloo_comm(engine: Engine, dist):
    """A communicator for libsnk over a gloo process group: the step's exchanges leave the device through the host
    (`snk_comm_create_callbacks`; the callbacks get the step's DEVICE pointers).  Not a production transport -- it is how the
    multi-process step (separate processes, separate contexts, real message passing) is run where RCCL cannot be: several ranks
    sharing ONE GPU in the tests and `bench.py --transport gloo`.  Returns (handle, keepalive)."""
    lib = engine.lib
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", engine.device)

    def dview(ptr, nbytes):
        return torch.as_tensor(_DevBytes(ptr, nbytes), device=dev)

    def a2a(_user, send, sbeg, scnt, recv, rbeg, rcnt, W):
        try:
            torch.cuda.synchronize(dev)                        # the step's kernels wrote `send` on its own streams
            s_hi = max([sbeg[p] + scnt[p] for p in range(W)] + [0])
            r_hi = max([rbeg[p] + rcnt[p] for p in range(W)] + [0])
            s_t = dview(send, s_hi) if s_hi else None
            r_t = dview(recv, r_hi) if r_hi else None
            if scnt[rank]:
                r_t[rbeg[rank]:rbeg[rank] + rcnt[rank]].copy_(s_t[sbeg[rank]:sbeg[rank] + scnt[rank]])
            ops, landing = [], []
            for q in range(W):
                if q == rank:
                    continue
                if scnt[q]:
                    ops.append(dist.P2POp(dist.isend, s_t[sbeg[q]:sbeg[q] + scnt[q]].cpu(), q))
                if rcnt[q]:
                    h = torch.empty(int(rcnt[q]), dtype=torch.uint8)
                    landing.append((q, h))
                    ops.append(dist.P2POp(dist.irecv, h, q))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for q, h in landing:
                r_t[rbeg[q]:rbeg[q] + rcnt[q]].copy_(h)
            torch.cuda.synchronize(dev)
            return 0
        except BaseException:      # a ctypes callback must not raise
            import traceback
            traceback.print_exc()
            return 1

    def gather(_user, mine, k, allp, W):
        try:
            m = torch.from_numpy(np.ctypeslib.as_array(C.cast(mine, C.POINTER(C.c_int64)), shape=(int(k),))).clone()      # host memory
            outs = [torch.empty_like(m) for _ in range(W)]
            dist.all_gather(outs, m)
            dst = torch.from_numpy(np.ctypeslib.as_array(C.cast(allp, C.POINTER(C.c_int64)), shape=(int(k) * int(W),)))
            for q in range(W):
                dst[q * k:(q + 1) * k].copy_(outs[q])
            return 0
        except BaseException:
            import traceback
            traceback.print_exc()
            return 1

    cb_a, cb_g = _lib.COMM_A2A(a2a), _lib.COMM_GATHER(gather)
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = lib.snk_comm_create_callbacks(rank, world, cb_a, cb_g, None, C.byref(h), err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    return int(h.value), (cb_a, cb_g)


class ShardedEngine:
    def __init__(self, engine: Engine, dist_or_comm, join: str | None = None):
        """dist_or_comm: torch.distributed (an initialised process group: one process per GPU, RCCL) or a communicator handle
        (local_world).  The join is owner-side: every rank writes the unitigs whose head fragment it owns."""
        self.eng = engine
        self.lib = engine.lib
        self._keep = None
        if isinstance(dist_or_comm, int):
            self.comm = dist_or_comm
        elif dist_or_comm.get_backend() == "gloo":
            self.comm, self._keep = gloo_comm(engine, dist_or_comm)      # ranks that share a GPU (tests, bench.py --transport gloo)
        else:
            self.comm = rccl_comm(engine, dist_or_comm)
        self.rank = int(self.lib.snk_comm_rank(self.comm))
        self.world = int(self.lib.snk_comm_world(self.comm))
        self.kind = self.lib.snk_comm_kind(self.comm).decode()

    def close(self):
        if getattr(self, "comm", None):
            self.lib.snk_comm_destroy(self.comm)
            self.comm = None

    def abort(self):
        if getattr(self, "comm", None):
            self.lib.snk_comm_abort(self.comm)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count_graph(self, rows, read_len, quals=None, bc=None, lens=None, good_len=None, params: Params | None = None,
                    ign_bc_below: int = 0, read_index_base: int = 0, total_reads: int = 0) -> ShardedResult:
        e, lib = self.eng, self.lib
        params = params or Params()
        err = C.create_string_buffer(512)
        r = _lib.SnkDevReads()
        r.n_reads, r.rows, r.row_words, r.read_len = rows.shape[0], rows.data_ptr(), rows.shape[1], read_len
        if lens is not None:
            r.lens = lens.data_ptr()
        if quals is not None:
            r.quals, r.qstride = quals.data_ptr(), quals.shape[1]
        if good_len is not None:
            r.good_len = good_len.data_ptr()
        if bc is not None:
            r.bc = bc.data_ptr()
        r.ign_bc_below, r.read_index_base = ign_bc_below, read_index_base
        p = params.to_c()
        raw = _lib.SnkShardResult()
        rc = lib.snk_shard_step(e._ctx, self.comm, C.byref(r), C.byref(p), int(total_reads), 0, C.byref(raw), e._stream(), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        return ShardedResult(e, params.K, raw)

    def count_graph_reads(self, r: "_lib.SnkDevReads", params: Params | None = None, total_reads: int = 0) -> ShardedResult:
        """The same step for reads described by plain device pointers (e.g. a rank's share of the ASSEMBLER_DF stage inputs in the compact
        form of snk_dev_ingest_df_trimmed: rows, good lengths, barcode ids; r.read_index_base = global index of the rank's first read)."""
        e, lib = self.eng, self.lib
        params = params or Params()
        err = C.create_string_buffer(512)
        p = params.to_c()
        raw = _lib.SnkShardResult()
        rc = lib.snk_shard_step(e._ctx, self.comm, C.byref(r), C.byref(p), int(total_reads), 0, C.byref(raw), e._stream(), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        return ShardedResult(e, params.K, raw)

    def count_graph_streamed(self, slabs, read_len, total_reads: int, rank_reads_ub: int, params: Params | None = None, ign_bc_below: int = 0) -> ShardedResult:
        """The same step with this rank's reads arriving slab by slab (snk_shard_stream_*): `slabs` yields dicts with rows, quals / good_len,
        optional lens, bc, and read_index_base (global index of the slab's first read); total_reads = reads of the whole job (every rank
        passes the same figure), rank_reads_ub = an upper bound of what this rank appends."""
        e, lib = self.eng, self.lib
        params = params or Params()
        err = C.create_string_buffer(512)
        p = params.to_c()
        it = iter(slabs)
        first = next(it)
        has_bc = 1 if first.get("bc") is not None else 0
        rc = lib.snk_shard_stream_begin(e._ctx, self.comm, C.byref(p), int(read_len), int(rank_reads_ub), int(total_reads), has_bc, e._stream(), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        import itertools
        keep = []
        for sl in itertools.chain([first], it):
            r = _lib.SnkDevReads()
            rows = sl["rows"]
            r.n_reads, r.rows, r.row_words, r.read_len = rows.shape[0], rows.data_ptr(), rows.shape[1], read_len
            if sl.get("lens") is not None:
                r.lens = sl["lens"].data_ptr()
            if sl.get("quals") is not None:
                r.quals, r.qstride = sl["quals"].data_ptr(), sl["quals"].shape[1]
            if sl.get("good_len") is not None:
                r.good_len = sl["good_len"].data_ptr()
            if sl.get("bc") is not None:
                r.bc = sl["bc"].data_ptr()
            r.ign_bc_below, r.read_index_base = ign_bc_below, int(sl.get("read_index_base", 0))
            rc = lib.snk_shard_stream_append(e._ctx, C.byref(r), e._stream(), err, 512)
            if rc != 0:
                raise _lib.SnkError(rc, err.value.decode(errors="replace"))
            keep.append(sl)          # (the launches read the slab's tensors: they stay alive until the step is through)
        raw = _lib.SnkShardResult()
        rc = lib.snk_shard_stream_finish(e._ctx, self.comm, 0, C.byref(raw), e._stream(), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        return ShardedResult(e, params.K, raw)
