"""Minimiser-sharded multi-GPU host logic: one process per GPU, torch.distributed for the
exchanges (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests of the plumbing).

The k-mer space is cut into NB_total minimiser buckets, rank r owns a contiguous range.  Per step:
  all-to-all #1  bucket histograms (u32 per bucket)            -> segment offsets on the owner
  all-to-all #2  supermer records (32 B each)                  -> every k-mer instance meets its owner
  all-to-all #3  membership queries for cross-rank neighbours  (24 B each, ~0.15 per retained k-mer)
  all-to-all #4  answers (4 B each)
  all-to-all #5/#6  fragment-link queries (24 B per fragment end with a remote neighbour) and answers (4 B)
  gather         fragments (k-mers, links, bases) -> rank 0, which ranks and writes them (tada's MAIN_ASM_SN)
Reference counterpart: the shardio exchange files + SHARD_ASM chunks + MAIN_ASM_SN of lib/tada
(rust-shardio/src/shard.rs:184-211,488-493; cmd_shard_asm.rs:37-94; cmd_main_asm.rs:25-89).
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


class SimWorld:
    """W in-process ranks (threads) that exchange through shared lists -- the SPMD code path of the real
    multi-GPU run on ONE GPU (tests), with the transport replaced by tensor copies."""

    def __init__(self, world: int, serial: bool = False):
        """serial=True: between two exchanges the ranks compute one after the other (rank 0 first), so the per-rank phase
        timings are those of a rank that has the GPU to itself (tools/sim_scale.py)."""
        self.world = world
        self.barrier_obj = threading.Barrier(world)
        self.slots = [None] * world
        self.serial = serial
        self.turn = threading.Condition()
        self.turn_of = 0

    def comm(self, rank: int) -> "SimComm":
        return SimComm(self, rank)


class SimComm:
    def __init__(self, w: SimWorld, rank: int):
        self.w, self.rank, self.world = w, rank, w.world

    def _end_section(self):       # my compute section is over: the next rank may start its own
        if self.w.serial:
            torch.cuda.synchronize()
            with self.w.turn:
                while self.w.turn_of != self.rank:
                    self.w.turn.wait()
                self.w.turn_of = self.rank + 1
                self.w.turn.notify_all()

    def _begin_section(self):     # after an exchange: wait until every lower rank has finished its section
        if self.w.serial:
            with self.w.turn:
                while self.w.turn_of != self.rank:
                    self.w.turn.wait()

    def _exchange(self, obj):
        self._end_section()
        self.w.slots[self.rank] = obj
        self.w.barrier_obj.wait()
        allv = list(self.w.slots)
        if self.w.serial and self.rank == 0:
            self.w.turn_of = 0
        self.w.barrier_obj.wait()
        return allv

    def allreduce_sum_int(self, v, device):
        r = sum(self._exchange(int(v)))
        self._begin_section()
        return r

    def all_gather_int(self, v, device):
        r = [int(x) for x in self._exchange(int(v))]
        self._begin_section()
        return r

    def all_to_all_v(self, send, send_counts, alloc=None):
        torch.cuda.synchronize()
        allv = self._exchange((send, list(send_counts)))
        parts, counts = [], []
        for (t, sc) in allv:
            off = sum(sc[: self.rank])
            parts.append(t[off: off + sc[self.rank]])
            counts.append(sc[self.rank])
        recv = torch.cat(parts) if parts else send[:0]
        torch.cuda.synchronize()
        self.w.barrier_obj.wait()     # nobody reuses its send buffer before everyone has copied
        self._begin_section()
        return recv, counts

    def exchange_ranged(self, send, recv, soff, roff, n_ranges):
        torch.cuda.synchronize()
        allv = self._exchange((send, soff))
        for src, (t, so) in enumerate(allv):
            a, b = so[self.rank][0], so[self.rank][n_ranges]
            if b > a:
                recv[roff[src][0]: roff[src][n_ranges]].copy_(t[a:b])
        torch.cuda.synchronize()
        self.w.barrier_obj.wait()     # nobody reuses its send buffer before everyone has copied
        self._begin_section()
        return [[] for _ in range(n_ranges)]

    def all_gather_v(self, send, counts, alloc=None):
        torch.cuda.synchronize()
        allv = self._exchange(send[:counts[self.rank]])
        out = torch.cat(list(allv)) if allv else send[:0]
        torch.cuda.synchronize()
        self.w.barrier_obj.wait()
        self._begin_section()
        return out

    def all_to_all_equal(self, send):
        m = send.shape[1]
        recv, _ = self.all_to_all_v(send.contiguous().view(-1).view(torch.uint8),
                                    [m * send.element_size()] * self.world)
        return recv.view(send.dtype).view(self.world, m)

    def barrier(self):
        self._exchange(None)
        self._begin_section()


# ------------------------------------------------------------------------------------------------ planning helpers
def plan_buckets(total_inst_upper: int, world: int, K: int) -> int:
    target = int(os.environ.get("SNK_TARGET_INST", "5000" if K == 48 else "3500"))
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
class ShardedResult:
    def __init__(self, eng: Engine, K: int):
        self._e = eng
        self.K = K
        self.phase_ms = {}
        self.kernel_ms = {}

    def _dl(self, ptr, nbytes, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        if nbytes:
            self._e._download(ptr, out.ctypes.data, nbytes)
        return out

    def keys(self):
        lohi = self._dl(self.frags.keys, self.n_kmers * 16, np.uint64, (self.n_kmers, 2))
        lo, hi = lohi[:, 0], lohi[:, 1]
        w = np.empty((self.n_kmers, 4), dtype=np.uint32)
        w[:, 0] = hi >> np.uint64(32); w[:, 1] = hi & np.uint64(0xFFFFFFFF)
        w[:, 2] = lo >> np.uint64(32); w[:, 3] = lo & np.uint64(0xFFFFFFFF)
        return w

    def counts(self):
        return self._dl(self.frags.counts, self.n_kmers * 4, np.uint32, (self.n_kmers,))

    def ctx(self):
        return self._dl(self.frags.ctx, self.n_kmers, np.uint8, (self.n_kmers,))

    def spectrum(self):
        nb = int(self.frags.spectrum_bins)
        return self._dl(self.frags.spectrum, nb * 8, np.uint64, (nb,))

    def unitigs(self) -> list[str]:
        """Canonical unitigs this rank wrote, sorted by BVComp: with the owner-side join every rank holds the unitigs whose
        head fragment it owns (their union over the ranks is the data set's unitig set); with join="rank0" rank 0 holds all."""
        u = self.joined
        if u is None:
            return []
        off = self._dl(u.unitig_off, (u.n_unitigs + 1) * 8, np.uint64, (u.n_unitigs + 1,))
        bases = self._dl(u.unitig_bases, u.total_bases, np.uint8, (u.total_bases,))
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        out = [lut[bases[int(off[i]):int(off[i + 1])]].tobytes().decode() for i in range(u.n_unitigs)]
        out.sort(key=lambda s: (-len(s), s))
        return out


class _BufferPool:
    """Grow-only named byte buffers reused across steps: the send / receive / query buffers are tens of GB, and a fresh
    torch.empty of that size is a device allocation (milliseconds each) whenever the caching allocator has no block of
    the right size left."""

    def __init__(self, device):
        self.device = device
        self.bufs: dict[str, torch.Tensor] = {}

    def get(self, name: str, nbytes: int) -> torch.Tensor:
        b = self.bufs.get(name)
        if b is None or b.numel() < nbytes:
            self.bufs.pop(name, None)
            b = torch.empty(int(nbytes * 1.1) + 4096, dtype=torch.uint8, device=self.device)
            self.bufs[name] = b
        return b[:nbytes]


class ShardedEngine:
    def __init__(self, engine: Engine, dist_or_comm, join: str | None = None):
        """join = "owner" (default): every rank writes the unitigs whose head fragment it owns -- no rank holds the whole job;
        "rank0": the first version, every fragment travels to rank 0 (tada's single-process MAIN_ASM_SN); SNK_JOIN overrides."""
        self.eng = engine
        self.comm = dist_or_comm if hasattr(dist_or_comm, "all_to_all_v") else TorchComm(dist_or_comm)
        self.pool = None
        self.join = join or os.environ.get("SNK_JOIN", "owner")
        assert self.join in ("owner", "rank0")
        # owner-side join: "partitioned" = every rank walks 1/W of the ranking (lists that are circles fall back to "replicated")
        self.rank_mode = os.environ.get("SNK_JOIN_RANK", "partitioned")
        assert self.rank_mode in ("partitioned", "replicated")

    def count_graph(self, rows, read_len, quals=None, bc=None, lens=None, good_len=None, params: Params | None = None,
                    ign_bc_below: int = 0, read_index_base: int = 0) -> ShardedResult:
        e, comm, lib = self.eng, self.comm, self.eng.lib
        W, me = comm.world, comm.rank
        params = params or Params()
        K = params.K
        dev = rows.device
        st = e._stream()
        err = C.create_string_buffer(512)
        if self.pool is None or self.pool.device != dev:
            self.pool = _BufferPool(dev)
        pool = self.pool

        def chk(rc):
            if rc != 0:
                raise _lib.SnkError(rc, err.value.decode(errors="replace"))

        ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        ev[0].record()
        marks = []          # (name, event): finer breakdown of the join (compute sections vs exchanges), res.join_ms

        def tick(name):
            m_ev = torch.cuda.Event(enable_timing=True)
            m_ev.record()
            marks.append((name, m_ev))
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

        inst_ub = comm.allreduce_sum_int(rows.shape[0] * max(0, read_len - K + 1), dev)
        NB_total = params.n_buckets if params.n_buckets else plan_buckets(inst_ub, W, K)
        NB_total = max(NB_total, (inst_ub >> 20) + 1)      # at most ~1 M instances per bucket (one workgroup counts a bucket)
        NB_total = -(-NB_total // W) * W
        NBl = NB_total // W
        # ---- stage 1: trim + histogram
        hist = torch.empty(NB_total, dtype=torch.int32, device=dev)
        ninst = C.c_uint64(0)
        chk(lib.snk_shard_hist(e._ctx, C.byref(r), C.byref(p), me, W, NB_total, hist.data_ptr(), C.byref(ninst), st, err, 512))
        ev[1].record()
        off64 = torch.zeros(NB_total + 1, dtype=torch.int64, device=dev)
        off64[1:] = torch.cumsum(hist.to(torch.int64), 0)
        n_super = int(off64[-1].item())
        if n_super >= (1 << 32):
            raise _lib.SnkError(-6, "more than 2^32 supermers on one rank")
        # the kernel reads u32; int32 storage holds the same bit patterns (values >= 2^31 wrap)
        offsets = torch.where(off64 >= (1 << 31), off64 - (1 << 32), off64).to(torch.int32)
        send = pool.get("send", max(n_super, 1) * 32)
        chk(lib.snk_shard_scatter(e._ctx, offsets.data_ptr(), send.data_ptr(), st, err, 512))
        ev[2].record()
        # ---- exchange #1/#2: histograms and records
        send_counts = owner_record_counts(off64, W)
        hist_recv = comm.all_to_all_equal(hist.view(W, NBl))
        nk = C.c_uint64(0)
        R = int(os.environ.get("SNK_EXCHANGE_RANGES", "4" if W > 1 else "1"))
        if R > 1 and hasattr(comm, "exchange_ranged"):
            # the records travel in R bucket ranges and range r is counted while range r+1 is still on the wire: both
            # sides know every piece size from the histograms, so nothing but the records is exchanged
            recv_counts = [int(x) for x in hist_recv.to(torch.int64).sum(dim=1).tolist()]
            seg_off = segment_offsets(hist_recv, recv_counts)
            recv = pool.get("recv", max(sum(recv_counts), 1) * 32)
            bounds = [NBl * q // R for q in range(R + 1)]
            bidx = torch.tensor(bounds, dtype=torch.int64, device=dev)
            soff = (off64[(torch.arange(W, device=dev, dtype=torch.int64) * NBl)[:, None] + bidx[None, :]] * 32).tolist()
            roff = (seg_off[:, bidx] * 32).tolist()
            works = comm.exchange_ranged(send, recv, soff, roff, R)
            ev[3].record()
            failure = []

            def ready(_user, q):
                try:
                    for wk in works[q]:
                        wk.wait()
                    return 0
                except BaseException as ex:          # a ctypes callback must not raise
                    failure.append(ex)
                    return 1
            cb = _lib.RANGE_READY(ready)
            barr = (C.c_uint32 * (R + 1))(*bounds)
            rc = lib.snk_shard_count_ranged(e._ctx, recv.data_ptr(), seg_off.data_ptr(), inst_ub // W, 1 if bc is not None else 0,
                                            R, barr, cb, None, C.byref(nk), st, err, 512)
            if failure:
                raise failure[0]
            chk(rc)
        else:
            recv, recv_bytes = comm.all_to_all_v(send[: n_super * 32], [c * 32 for c in send_counts], alloc=lambda nb: pool.get("recv", nb))
            seg_off = segment_offsets(hist_recv, [b // 32 for b in recv_bytes])
            ev[3].record()
            # ---- stage 3: count
            chk(lib.snk_shard_count(e._ctx, recv.data_ptr(), seg_off.data_ptr(), inst_ub // W, 1 if bc is not None else 0,
                                    C.byref(nk), st, err, 512))
        ev[4].record()
        # ---- stage 4: prune with remote queries
        qcount = (C.c_uint64 * W)()
        chk(lib.snk_shard_prune_plan(e._ctx, qcount, st, err, 512))
        qc = [int(x) for x in qcount]
        qoff = torch.zeros(W + 1, dtype=torch.int64, device=dev)
        qoff[1:] = torch.cumsum(torch.tensor(qc, dtype=torch.int64, device=dev), 0)
        nq = sum(qc)
        qbuf = pool.get("qbuf", max(nq, 1) * 24)
        chk(lib.snk_shard_prune_fill(e._ctx, qoff.data_ptr(), qbuf.data_ptr(), st, err, 512))
        qin, qin_bytes = comm.all_to_all_v(qbuf[: nq * 24], [c * 24 for c in qc], alloc=lambda nb: pool.get("qin", nb))
        nq_in = qin.numel() // 24
        ans = pool.get("ans", max(nq_in, 1) * 4)
        chk(lib.snk_shard_prune_answer(e._ctx, qin.data_ptr(), nq_in, ans.data_ptr(), st, err, 512))
        ans_back, _ = comm.all_to_all_v(ans[: nq_in * 4], [b // 24 * 4 for b in qin_bytes], alloc=lambda nb: pool.get("ans_back", nb))
        assert ans_back.numel() == nq * 4
        chk(lib.snk_shard_prune_apply(e._ctx, qbuf.data_ptr(), ans_back.data_ptr(), nq, qoff.data_ptr(), st, err, 512))
        ev[5].record()
        # ---- stage 5: local fragments
        all_n = comm.all_gather_int(int(nk.value), dev)
        node_off = torch.zeros(W + 1, dtype=torch.int64, device=dev)
        node_off[1:] = torch.cumsum(torch.tensor(all_n, dtype=torch.int64, device=dev), 0)
        fr = _lib.SnkShardFrags()
        chk(lib.snk_shard_fragments(e._ctx, node_off.data_ptr(), int(node_off[me].item()), C.byref(fr), st, err, 512))
        ev[6].record()
        # ---- gather fragments on rank 0 and join
        res = ShardedResult(e, K)
        res.frags = fr
        res.n_kmers = int(fr.n_kmers)
        res.n_instances = int(ninst.value)
        res.n_supermers = n_super
        res.n_buckets = NB_total
        res.n_queries = nq
        res.n_frags = int(fr.n_frags)
        F = int(fr.n_frags)

        def dcopy(name, ptr, nbytes):
            """device-to-device copy of a library-owned buffer into a pooled torch tensor (send buffer)."""
            t = pool.get(name, max(nbytes, 8))
            if nbytes:
                torch.cuda.current_stream().synchronize()
                _copy_d2d(t.data_ptr(), ptr, nbytes)
            return t[:nbytes]

        # ---- links between fragments, decided on the owners: an end asks the rank that owns the state it points at
        tick("start")
        all_F = comm.all_gather_int(F, dev)
        frag_off = [0]
        for x in all_F:
            frag_off.append(frag_off[-1] + x)
        if 2 * frag_off[-1] >= (1 << 32):
            raise _lib.SnkError(-6, "more than 2^31 fragments in the job")
        lq = (C.c_uint64 * W)()
        chk(lib.snk_shard_links_plan(e._ctx, frag_off[me], lq, st, err, 512))
        lqc = [int(x) for x in lq]
        lqoff = torch.zeros(W + 1, dtype=torch.int64, device=dev)
        lqoff[1:] = torch.cumsum(torch.tensor(lqc, dtype=torch.int64, device=dev), 0)
        nlq = sum(lqc)
        lqbuf = pool.get("lqbuf", max(nlq, 1) * 24)
        chk(lib.snk_shard_links_fill(e._ctx, lqoff.data_ptr(), lqbuf.data_ptr(), st, err, 512))
        lqin, lqin_bytes = comm.all_to_all_v(lqbuf[: nlq * 24], [c * 24 for c in lqc], alloc=lambda nb: pool.get("lqin", nb))
        nlq_in = lqin.numel() // 24
        lans = pool.get("lans", max(nlq_in, 1) * 4)
        chk(lib.snk_shard_links_answer(e._ctx, lqin.data_ptr(), nlq_in, lans.data_ptr(), st, err, 512))
        lans_back, _ = comm.all_to_all_v(lans[: nlq_in * 4], [b // 24 * 4 for b in lqin_bytes], alloc=lambda nb: pool.get("lans_back", nb))
        assert lans_back.numel() == nlq * 4
        flink_p = C.c_void_p()
        chk(lib.snk_shard_links_apply(e._ctx, lqbuf.data_ptr(), lans_back.data_ptr(), nlq, C.byref(flink_p), st, err, 512))
        res.n_link_queries = nlq
        tick("links(x)")
        if self.join == "owner":
            # ---- owner-side join: every rank sees the job's LINK structure only (12 bytes per fragment), ranks the fragment
            # lists, places its own fragments and sends each to the rank that owns its unitig's head, which writes the unitig
            Ft = frag_off[-1]
            nk_all = comm.all_gather_v(dcopy("s_nk", fr.nk, F * 4), [x * 4 for x in all_F], alloc=lambda nb: pool.get("g_nk", max(nb, 8))[:nb])
            fl_all = comm.all_gather_v(dcopy("s_link", flink_p.value, F * 8), [x * 8 for x in all_F], alloc=lambda nb: pool.get("g_link", max(nb, 8))[:nb])
            d_frag_off = torch.tensor(frag_off, dtype=torch.int64, device=dev)
            tick("gather_links(x)")
            fto, bto = (C.c_uint64 * W)(), (C.c_uint64 * W)()
            ranked = False
            if self.rank_mode == "partitioned":
                # the two walks of the ruling-set ranking for a 1/W share of the splitters on every rank; 16 B per splitter
                # (all-gather) and 16 B per state (to its owner) cross the links
                m_spl, w1p = C.c_uint64(0), C.c_void_p()
                chk(lib.snk_shard_prank_begin(e._ctx, Ft, nk_all.data_ptr(), fl_all.data_ptr(), frag_off[me], C.byref(m_spl), C.byref(w1p), st, err, 512))
                m = int(m_spl.value)
                tick("rank_setup+walk1")
                shares = [(m * (q + 1) // W - m * q // W) * 16 for q in range(W)]
                w1_all = comm.all_gather_v(dcopy("s_w1", w1p.value, shares[me]), shares, alloc=lambda nb: pool.get("g_w1", max(nb, 16))[:nb])
                tick("gather_splitters(x)")
                rto, circ = (C.c_uint64 * W)(), C.c_uint32(0)
                chk(lib.snk_shard_prank_walk(e._ctx, w1_all.data_ptr(), d_frag_off.data_ptr(), rto, C.byref(circ), st, err, 512))
                if not circ.value:
                    rto = [int(x) for x in rto]
                    roff_ = [0]
                    for q in range(W):
                        roff_.append(roff_[-1] + rto[q])
                    d_roff = torch.tensor(roff_[:W], dtype=torch.int64, device=dev)
                    rsend = pool.get("s_rk", max(roff_[-1], 1) * 16)
                    chk(lib.snk_shard_prank_route(e._ctx, d_frag_off.data_ptr(), d_roff.data_ptr(), rsend.data_ptr(), st, err, 512))
                    torch.cuda.current_stream().synchronize()
                    tick("jump+walk2+route")
                    rk_in, _ = comm.all_to_all_v(rsend[: roff_[-1] * 16], [c * 16 for c in rto], alloc=lambda nb: pool.get("g_rk", max(nb, 16))[:nb])
                    tick("ranks_to_owners(x)")
                    chk(lib.snk_shard_place_ranked(e._ctx, K, rk_in.data_ptr(), rk_in.numel() // 16, d_frag_off.data_ptr(), fto, bto, st, err, 512))
                    ranked = True
                    res.exchange_bytes_rank = (m * 16, roff_[-1] * 16)
            if not ranked:      # a list is a circle (same verdict on every rank: it comes from replicated data), or rank_mode == "replicated"
                chk(lib.snk_shard_place(e._ctx, K, Ft, nk_all.data_ptr(), fl_all.data_ptr(), d_frag_off.data_ptr(), frag_off[me], fto, bto, st, err, 512))
            res.join_ranking = "partitioned" if ranked else "replicated"
            tick("place")
            fto, bto = [int(x) for x in fto], [int(x) for x in bto]
            hoff, boff_, bpad = route_offsets(fto, bto)
            d_hoff = torch.tensor(hoff[:W], dtype=torch.int64, device=dev)
            d_boff = torch.tensor(boff_[:W], dtype=torch.int64, device=dev)
            hdr = pool.get("s_hdr", max(hoff[-1], 1) * 32)
            sb = pool.get("s_bases", max(boff_[-1], 16))
            chk(lib.snk_shard_route_fill(e._ctx, K, d_frag_off.data_ptr(), d_hoff.data_ptr(), d_boff.data_ptr(), hdr.data_ptr(), sb.data_ptr(), st, err, 512))
            torch.cuda.current_stream().synchronize()
            tick("route_fill")
            hdr_in, hdr_bytes = comm.all_to_all_v(hdr[: hoff[-1] * 32], [c * 32 for c in fto], alloc=lambda nb: pool.get("g_hdr", max(nb, 8))[:nb])
            b_in, b_bytes = comm.all_to_all_v(sb[: boff_[-1]], bpad, alloc=lambda nb: pool.get("g_bases", max(nb, 16))[:nb])
            hseg, bseg = recv_segments(hdr_bytes, b_bytes)
            tick("fragments_to_owners(x)")
            d_hseg = torch.tensor(hseg, dtype=torch.int64, device=dev)
            d_bseg = torch.tensor(bseg, dtype=torch.int64, device=dev)
            un = _lib.SnkShardUnitigs()
            chk(lib.snk_shard_emit(e._ctx, K, hseg[-1], hdr_in.data_ptr(), d_hseg.data_ptr(), d_bseg.data_ptr(), b_in.data_ptr(), C.byref(un), st, err, 512))
            tick("emit")
            res.joined = un
            res.n_unitigs = int(un.n_unitigs)
            res.exchange_bytes_join = (Ft * 12, hoff[-1] * 32 + boff_[-1])
            res._keep = (nk_all, fl_all, hdr_in, b_in, d_hseg, d_bseg, d_frag_off)
        else:
            # ---- gather on rank 0: k-mers per fragment, links, starts, bases
            to0 = lambda n: [n if q == 0 else 0 for q in range(W)]
            TB = int(fr.total_bases)
            t_nk, _ = comm.all_to_all_v(dcopy("s_nk", fr.nk, F * 4), to0(F * 4), alloc=lambda nb: pool.get("g_nk", nb))
            t_link, _ = comm.all_to_all_v(dcopy("s_link", flink_p.value, F * 8), to0(F * 8), alloc=lambda nb: pool.get("g_link", nb))
            t_start, start_bytes = comm.all_to_all_v(dcopy("s_start", fr.boff, F * 8), to0(F * 8), alloc=lambda nb: pool.get("g_start", nb))
            if W == 1:
                t_bases, base_bytes = comm.all_to_all_v(dcopy("s_bases", fr.bases, TB), to0(TB), alloc=lambda nb: pool.get("g_bases", nb))
            else:
                # the bases are the bulk of the gather (1 byte per k-mer + 47 per fragment): they cross xGMI at 2 bits each
                all_TB = comm.all_gather_int(TB, dev)
                pb = int(lib.snk_pack2_bytes(TB))
                t_pk = pool.get("s_bases2", max(pb, 8))
                if TB:
                    chk(lib.snk_dev_pack2(e._ctx, fr.bases, TB, t_pk.data_ptr(), st))
                    torch.cuda.current_stream().synchronize()
                t_pk_all, pk_bytes = comm.all_to_all_v(t_pk[:pb], to0(pb), alloc=lambda nb: pool.get("g_bases2", nb))
                base_bytes = [0] * W
                t_bases = t_pk_all[:0]
                if me == 0:
                    base_bytes = [((x + 15) // 16) * 16 for x in all_TB]          # every rank's bases start 16-byte aligned
                    t_bases = pool.get("g_bases", max(sum(base_bytes), 16))[:sum(base_bytes)]
                    a = b = 0
                    for q in range(W):
                        if all_TB[q]:
                            chk(lib.snk_dev_unpack2(e._ctx, t_pk_all.data_ptr() + a, all_TB[q], t_bases.data_ptr() + b, st))
                        a += pk_bytes[q]
                        b += base_bytes[q]
            res.joined = None
            res.n_unitigs = 0
            if me == 0:
                Ft = t_nk.numel() // 4
                # fragment starts are offsets into their rank's base buffer: shift by the buffers in front
                # (in place, one slice per source rank: torch.repeat_interleave over the fragments took 6.8 ms per 17.5 M)
                starts_all = t_start.view(torch.int64)
                a, shift = 0, 0
                for q in range(W):
                    nq = start_bytes[q] // 8
                    if shift and nq:
                        starts_all[a:a + nq] += shift
                    a += nq
                    shift += base_bytes[q]
                if starts_all.numel() == 0:
                    starts_all = torch.zeros(1, dtype=torch.int64, device=dev)
                if t_link.numel() == 0:
                    t_link = torch.zeros(8, dtype=torch.uint8, device=dev)
                un = _lib.SnkShardUnitigs()
                chk(lib.snk_shard_join_linked(e._ctx, K, Ft, t_nk.data_ptr(), None, None, t_link.data_ptr(), starts_all.data_ptr(),
                                              t_bases.data_ptr(), t_bases.numel(), C.byref(un), st, err, 512))
                res.joined = un
                res.n_unitigs = int(un.n_unitigs)
                res._keep = (t_nk, t_link, starts_all, t_bases)
        ev[7].record()
        torch.cuda.synchronize()
        names = ["partition", "compact", "exchange", "count", "prune", "fragments", "join"]
        res.phase_ms = {names[i]: ev[i].elapsed_time(ev[i + 1]) for i in range(7)}
        res.phase_ms["total"] = ev[0].elapsed_time(ev[7])
        # sections of the join; "(x)" = an exchange (transport + the waits for the other ranks), the rest is this rank's compute
        res.join_ms = {marks[i + 1][0]: marks[i][1].elapsed_time(marks[i + 1][1]) for i in range(len(marks) - 1)}
        res.kernel_ms = {"count": float(fr.count_kernel_ms)}
        res.buckets_split = int(fr.buckets_split)
        return res


def _copy_d2d(dst_ptr: int, src_ptr: int, nbytes: int) -> None:
    """hipMemcpy device-to-device through the process's HIP runtime (the one torch uses)."""
    hip = _hip()
    rc = hip.hipMemcpy(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_size_t(nbytes), 3)  # hipMemcpyDeviceToDevice = 3
    if rc != 0:
        raise RuntimeError(f"hipMemcpy D2D failed ({rc})")


_hip_handle = None


def _hip():
    global _hip_handle
    if _hip_handle is None:
        from pathlib import Path
        cand = Path(torch.__file__).parent / "lib" / "libamdhip64.so"
        _hip_handle = C.CDLL(str(cand if cand.exists() else "libamdhip64.so"), mode=C.RTLD_GLOBAL)
        _hip_handle.hipMemcpy.restype = C.c_int
        _hip_handle.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip_handle
