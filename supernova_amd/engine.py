"""Host-side engine: one process per GPU, device memory and streams through torch, compute in libsnk.

`Engine.count_graph` is the Python face of the C++ entry the reference exposes for this path,
buildReadQGraph48 (lib/assembly/src/paths/long/BuildReadQGraph48.h:24-34): same thresholds
(minQual, minFreq, minBC, ignBcBelow), same outputs (retained k-mer dictionary with pruned contexts,
canonical unitigs), with reads resident in HBM.  No CPU fallback: without libsnk.so or without a
gfx950 device every call raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import lib as _lib

PHASES = ("trim", "plan", "partition", "count", "sort", "graph", "-", "total")


@dataclass
class Params:
    K: int = 48
    min_qual: int = 7      # lib/tada/mro/_asm_sn.mro:15, 10X/DF.cc:141
    min_freq: int = 3      # mro/_assembler.mro:44, 10X/DF.cc:139
    min_bc: int = 2        # 10X/DF.cc:140
    n_buckets: int = 0
    graph: bool = True
    sorted_table: bool = True     # False: leave the retained table in bucket order (SNK_F_UNSORTED_TABLE)
    global_graph: bool = False    # True: the global graph stage (SNK_F_GLOBAL_GRAPH), a cross-check of the bucket-local one
    grouped: bool = False         # True: per-group graphs (SNK_F_GROUPED); count_graph(group=...) gives the group of every read
    long_minimiser: bool = False  # True: 20-base minimisers (SNK_F_LONG_MINIMISER): genomes of human size; same results

    def to_c(self) -> _lib.SnkParams:
        p = _lib.SnkParams()
        p.K, p.min_qual, p.min_freq, p.min_bc = self.K, self.min_qual, self.min_freq, self.min_bc
        p.n_buckets = self.n_buckets
        p.flags = ((0 if self.graph else 1) | (0 if self.sorted_table else 2) | (4 if self.global_graph else 0)
                   | (8 if self.grouped else 0) | (64 if self.long_minimiser else 0))
        return p


class Result:
    """Device-resident result of one count_graph call (valid until the next call on the same engine)."""

    def __init__(self, engine: "Engine", raw: _lib.SnkDevResult, K: int):
        self._e = engine
        self.raw = raw
        self.K = K
        for f in ("n_reads", "n_instances", "n_supermers", "n_buckets", "n_kmers", "n_unitigs", "unitig_total_bases",
                  "n_circles", "rank_rounds", "buckets_split", "max_slots_used", "scratch_bytes", "n_boundary", "n_overflow",
                  "n_fragments", "repartitioned", "n_hot_buckets"):
            setattr(self, f, int(getattr(raw, f)))
        self.phase_ms = {PHASES[i]: float(raw.phase_ms[i]) for i in range(8) if PHASES[i] != "-"}
        self.kernel_ms = {"partition": float(raw.kernel_ms[1]),
                          "count": float(raw.kernel_ms[2])}
        names = ("local_prune", "-", "fragments", "join", "table")     # local_prune includes the boundary index + resolve
        self.graph_ms = {names[i]: float(raw.graph_ms[i]) for i in range(5) if names[i] != "-"}

    def _dl(self, ptr, nbytes, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        if nbytes:
            self._e._download(ptr, out.ctypes.data, nbytes)
        return out

    def good_len(self) -> np.ndarray:
        return self._dl(self.raw.good_len, self.n_reads * 2, np.uint16, (self.n_reads,))

    def keys(self) -> np.ndarray:
        """[n_kmers, 4] u32 words MSB-first (include/snk.h key convention), ascending."""
        lohi = self._dl(self.raw.keys, self.n_kmers * 16, np.uint64, (self.n_kmers, 2))
        lo, hi = lohi[:, 0], lohi[:, 1]
        w = np.empty((self.n_kmers, 4), dtype=np.uint32)
        w[:, 0] = hi >> np.uint64(32)
        w[:, 1] = hi & np.uint64(0xFFFFFFFF)
        w[:, 2] = lo >> np.uint64(32)
        w[:, 3] = lo & np.uint64(0xFFFFFFFF)
        return w

    def counts(self) -> np.ndarray:
        return self._dl(self.raw.counts, self.n_kmers * 4, np.uint32, (self.n_kmers,))

    def ctx(self) -> np.ndarray:
        return self._dl(self.raw.ctx, self.n_kmers, np.uint8, (self.n_kmers,))

    def spectrum(self) -> np.ndarray:
        nb = int(self.raw.spectrum_bins)
        return self._dl(self.raw.spectrum, nb * 8, np.uint64, (nb,))

    def unitig_arrays(self):
        off = self._dl(self.raw.unitig_off, (self.n_unitigs + 1) * 8, np.uint64, (self.n_unitigs + 1,))
        bases = self._dl(self.raw.unitig_bases, self.unitig_total_bases, np.uint8, (self.unitig_total_bases,))
        return off, bases

    def unitig_groups(self) -> np.ndarray:
        """Grouped runs: group id of every unitig (same order as unitig_arrays())."""
        return self._dl(self.raw.unitig_group, self.n_unitigs * 4, np.uint32, (self.n_unitigs,))

    def bv_image_device(self):
        """a13 on the device: (device pointer, bytes) of the .bv hand-off file -- "BINWRITE", count, per unitig u32 length + 2-bit bases in
        BVComp order (snk_dev_bv_image).  Context memory: valid until the engine's next top-level call."""
        ptr, nb = C.c_void_p(0), C.c_uint64(0)
        err = C.create_string_buffer(512)
        rc = self._e.lib.snk_dev_bv_image(self._e._ctx, int(self.K), self.n_unitigs, self.raw.unitig_off, self.raw.unitig_bases, 1,
                                          C.byref(ptr), C.byref(nb), self._e._stream(), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        return ptr.value, int(nb.value)

    def bv_image(self) -> bytes:
        ptr, nb = self.bv_image_device()
        return self._dl(ptr, nb, np.uint8, (nb,)).tobytes()

    def hbv(self) -> dict:
        """The graph from the device-resident unitigs (buildHBVFromEdges, HBVFromEdges.cc:244-296; snk_dev_hbv):
        vertex ids per HBV edge, fwd/rev translation per unitig, numbered in BVComp order; 'order'[r] = index of the
        rank-r unitig in unitig_arrays().  Must be called before the engine's next count_graph."""
        h = _lib.SnkHbv()
        ms = C.c_float(0)
        err = C.create_string_buffer(512)
        rc = self._e.lib.snk_dev_hbv(self._e._ctx, int(self.K), self.n_unitigs, self.raw.unitig_off, self.raw.unitig_bases,
                                     C.byref(h), C.byref(ms), self._e._stream(), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        ne, nu = h.n_edges, self.n_unitigs
        arr = lambda p, m, dt: (np.array(np.ctypeslib.as_array(p, shape=(m,)), dtype=dt, copy=True) if m else np.zeros(0, dt))      # (one copy: the C arrays are freed below)
        out = dict(n_vertices=h.n_vertices, n_edges=ne, v_left=arr(h.v_left, ne, np.int32), v_right=arr(h.v_right, ne, np.int32),
                   src=arr(h.src_unitig, ne, np.int32), is_rc=arr(h.is_rc, ne, np.uint8), fwd=arr(h.fwd_xlat, nu, np.int32),
                   rev=arr(h.rev_xlat, nu, np.int32), order=arr(h.bvcomp_order, nu, np.int32), device_ms=float(ms.value))
        self._e.lib.snk_hbv_free(C.byref(h))
        return out

    def path_reads(self, rows, read_len: int, quals, lens=None, mark_dups=False, bc=None, unitig_bcs=False, download=True, bcs_nocut=False):
        """f1: the reads (untrimmed packed rows + quality rows on the device) onto the graph of this result's unitigs --
        pathReads with the new aligner (BuildReadQGraph48.cc:1441-1469).  Returns (offset i32[n], n_edges u32[n], edges i32[sum],
        info) on the host, HBV edge ids as numbered by buildHBVFromEdges.  Must be called before the engine's next count_graph.
        mark_dups: f4, MarkDups over these paths (10X/SecretOps.cc:413-593; bc = raw barcode ids on the device or None) ->
        info['dups'] = dict(dup u8[n/2], interdup_rate, n_dup_pairs, n_dup_reads, n_art_pairs, n_placed, ms).
        unitig_bcs: the rest of f4 -- per unitig (numbering of unitig_arrays()) the sorted distinct barcodes > 0 of the reads with a
        k-mer on it (tada's edge -> barcode sets) -> info['unitig_bcs'] = (off u64[U+1], bcs u32[...]); unitig_bcs="exhaustive" derives them
        the slow, literal way (every k-mer of every barcoded read looked up); bcs_nocut: without the 20 000-entry cut."""
        e = self._e
        h = _lib.SnkHbv()
        ms = C.c_float(0)
        err = C.create_string_buffer(512)
        rc = e.lib.snk_dev_hbv(e._ctx, int(self.K), self.n_unitigs, self.raw.unitig_off, self.raw.unitig_bases, C.byref(h), C.byref(ms),
                               e._stream(), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        try:
            r = _lib.SnkDevReads()
            r.n_reads, r.rows, r.row_words, r.read_len = rows.shape[0], rows.data_ptr(), rows.shape[1], read_len
            r.quals, r.qstride = quals.data_ptr(), quals.shape[1]
            if lens is not None:
                r.lens = lens.data_ptr()
            out = _lib.SnkDevPaths()
            if unitig_bcs and bc is not None:
                r.bc = bc.data_ptr()
            rc = e.lib.snk_dev_path_reads2(e._ctx, int(self.K), C.byref(r), self.n_unitigs, self.raw.unitig_off, self.raw.unitig_bases,
                                           C.byref(h), (0 if not unitig_bcs else 1 | (2 if unitig_bcs == "exhaustive" else 0) | (4 if bcs_nocut else 0)), C.byref(out), e._stream(), err, 512)
            if rc:
                raise _lib.SnkError(rc, err.value.decode(errors="replace"))
            dups = None
            if mark_dups:
                if bc is not None:
                    r.bc = bc.data_ptr()
                dd = _lib.SnkDevDups()
                rc = e.lib.snk_dev_mark_dups(e._ctx, C.byref(r), C.byref(out), C.byref(dd), e._stream(), err, 512)
                if rc:
                    raise _lib.SnkError(rc, err.value.decode(errors="replace"))
                npairs = int(dd.n_pairs)
                dups = dict(dup=self._dl(dd.dup, npairs, np.uint8, (npairs,)) if download else None, interdup_rate=float(dd.interdup_rate),
                            n_dup_pairs=int(dd.n_dup_pairs), n_dup_reads=int(dd.n_dup_reads), n_interdup_reads=int(dd.n_interdup_reads),
                            n_art_pairs=int(dd.n_art_pairs), n_placed=int(dd.n_placed), ms=float(dd.ms))
        finally:
            e.lib.snk_hbv_free(C.byref(h))
        n, tot = int(out.n_reads), int(out.n_edges_total)
        if not download:          # timings only (bench.py: the results stay on the device)
            info = dict(dict_ms=float(out.dict_ms), path_ms=float(out.path_ms), bcs_ms=float(out.bcs_ms), hbv_device_ms=float(ms.value),
                        dict_slots=int(out.dict_slots), n_edges_total=tot, n_unitig_bcs=int(out.n_unitig_bcs), n_slow=int(out.n_slow),
                        lookup=("index" if out.lookup_index else "kmer_dictionary"))
            if dups is not None:
                info["dups"] = {k: v for k, v in dups.items() if k != "dup"}
            return None, None, None, info
        off = self._dl(out.offset, n * 4, np.int32, (n,))
        ne = self._dl(out.n_edges, n * 4, np.uint32, (n,))
        edges = self._dl(out.edges, tot * 4, np.int32, (tot,))
        info = dict(dict_ms=float(out.dict_ms), path_ms=float(out.path_ms), bcs_ms=float(out.bcs_ms), hbv_device_ms=float(ms.value), dict_slots=int(out.dict_slots), lookup=("index" if out.lookup_index else "kmer_dictionary"),
                    n_slow=int(out.n_slow))
        if dups is not None:
            info["dups"] = dups
        if unitig_bcs and out.unitig_bc_off:
            nb = int(out.n_unitig_bcs)
            info["unitig_bcs"] = (self._dl(out.unitig_bc_off, (self.n_unitigs + 1) * 8, np.uint64, (self.n_unitigs + 1,)),
                                  self._dl(out.unitig_bcs, nb * 4, np.uint32, (nb,)))
        return off, ne, edges, info

    def unitigs(self) -> list[str]:
        """Canonical unitigs sorted by (length desc, lexicographic) = BVComp, HBVFromEdges.cc:106-111."""
        off, bases = self.unitig_arrays()
        asc = np.frombuffer(b"ACGT", dtype=np.uint8)[bases].tobytes().decode()
        us = [asc[int(off[i]):int(off[i + 1])] for i in range(self.n_unitigs)]
        us.sort(key=lambda s: (-len(s), s))
        return us


import weakref as _weakref

_LIVE = _weakref.WeakSet()      # contexts that are open (tests pin options on all of them: tests/conftest.py `tune`)


def live_engines():
    return [e for e in list(_LIVE) if getattr(e, "_ctx", None)]


class Engine:
    def __init__(self, device: int | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("supernova_amd.Engine needs a gfx950 GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.cuda.current_device() if device is None else device
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self.lib.snk_ctx_create(self.device, C.byref(h), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        self._ctx = h
        _LIVE.add(self)

    def last_count_limit(self) -> int:
        """Usable count-table slots per pass in the last count_graph call (1216, or 1920 with booked slots; snk_ctx_last_count_limit)."""
        return int(self.lib.snk_ctx_last_count_limit(self._ctx))

    def last_partition_passes(self) -> int:
        """Bucket-range passes of the last count_graph call (1 = the one-pass partition; snk_ctx_last_partition_passes)."""
        return int(self.lib.snk_ctx_last_partition_passes(self._ctx))

    # ---- tuning (include/snk.h "tuning"): options live in the context, not in the environment
    def set_option(self, name: str, value: int):
        err = C.create_string_buffer(512)
        rc = self.lib.snk_ctx_set_option(self._ctx, name.encode(), int(value), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))

    def clear_option(self, name: str | None = None):
        """Back to the library's own choice (None: every option)."""
        self.lib.snk_ctx_clear_option(self._ctx, name.encode() if name is not None else None)

    def get_option(self, name: str):
        """-> the pinned value, or None when the library chooses."""
        v = C.c_longlong(0)
        rc = self.lib.snk_ctx_get_option(self._ctx, name.encode(), C.byref(v))
        if rc < 0:
            raise KeyError(name)
        return int(v.value) if rc == 1 else None

    def options(self) -> dict:
        """name -> description of every option the library knows."""
        out, i = {}, 0
        while True:
            n = self.lib.snk_option_name(i)
            if not n:
                return out
            out[n.decode()] = self.lib.snk_option_doc(i).decode()
            i += 1

    def set_tuning(self, **fields):
        """snk_ctx_set_tuning: the documented knobs as one struct (count_kernel=2, target_inst=4000, ...); unnamed fields = the library's choice."""
        t = _lib.SnkTuning()
        self.lib.snk_tuning_default(C.byref(t))
        for k, v in fields.items():
            if not hasattr(t, k) or k.startswith("last_") or k == "reserved":
                raise AttributeError(k)
            setattr(t, k, int(v))
        err = C.create_string_buffer(512)
        rc = self.lib.snk_ctx_set_tuning(self._ctx, C.byref(t), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))

    def get_tuning(self) -> dict:
        """What is pinned (0 = the library chooses) and, in last_*, what the context's last call ran with."""
        t = _lib.SnkTuning()
        self.lib.snk_ctx_get_tuning(self._ctx, C.byref(t))
        return {k: int(getattr(t, k)) for k, _ in _lib.SnkTuning._fields_ if k != "reserved"}

    def reserve(self, n_bytes: int):
        """Map n_bytes of device memory into the context's scratch arena now and keep them mapped between calls (snk_ctx_reserve): a call
        that outgrows the arena otherwise pays the driver ~25-30 ms per new GB inside the call."""
        err = C.create_string_buffer(512)
        rc = self.lib.snk_ctx_reserve(self._ctx, int(n_bytes), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))

    def release_cache(self):
        """Hand the context's cached, unused device memory back (snk_ctx_trim): for callers that change problem size and share the GPU."""
        self.lib.snk_ctx_trim(self._ctx)

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.snk_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _download(self, dptr, hptr, nbytes):
        _lib.check(self.lib.snk_dev_download(self._ctx, dptr, hptr, nbytes, self._stream()))

    # ---- synthetic reads straight into HBM
    def synth(self, sp: _lib.SnkSynthParams, first: int = 0, n: int | None = None, qstride: int | None = None):
        n = sp.n_reads - first if n is None else n
        rw = (sp.read_len + 15) // 16
        qs = qstride or ((sp.read_len + 15) // 16 * 16)
        dev = torch.device("cuda", self.device)
        rows = torch.empty((n, rw), dtype=torch.int32, device=dev)
        quals = torch.empty((n, qs), dtype=torch.uint8, device=dev)
        bc = torch.empty((n,), dtype=torch.int32, device=dev)
        _lib.check(self.lib.snk_synth_dev(self._ctx, C.byref(sp), first, n, rows.data_ptr(), rw, quals.data_ptr(), qs,
                                          bc.data_ptr(), self._stream()))
        return rows, quals, bc

    def trim(self, quals: torch.Tensor, read_len: int, K: int = 48, min_qual: int = 7, lens: torch.Tensor | None = None):
        n, qs = quals.shape
        out = torch.empty((n,), dtype=torch.int16, device=quals.device)
        _lib.check(self.lib.snk_dev_trim(self._ctx, quals.data_ptr(), qs, lens.data_ptr() if lens is not None else None,
                                         read_len, n, K, min_qual, out.data_ptr(), self._stream()))
        return out

    def pack_ascii(self, ascii_rows: torch.Tensor, read_len: int):
        n, stride = ascii_rows.shape
        rw = (read_len + 15) // 16
        rows = torch.empty((n, rw), dtype=torch.int32, device=ascii_rows.device)
        _lib.check(self.lib.snk_dev_pack_ascii(self._ctx, ascii_rows.data_ptr(), stride, read_len, n, rows.data_ptr(), rw,
                                               self._stream()))
        return rows

    def count_graph(self, rows: torch.Tensor, read_len: int, quals: torch.Tensor | None = None,
                    bc: torch.Tensor | None = None, lens: torch.Tensor | None = None,
                    good_len: torch.Tensor | None = None, params: Params | None = None, ign_bc_below: int = 0,
                    read_index_base: int = 0, group: torch.Tensor | None = None) -> Result:
        params = params or Params()
        assert rows.is_cuda and rows.dtype == torch.int32 and rows.is_contiguous()
        r = _lib.SnkDevReads()
        r.n_reads = rows.shape[0]
        r.rows = rows.data_ptr()
        r.row_words = rows.shape[1]
        r.read_len = read_len
        if lens is not None:
            assert lens.dtype == torch.int16 and lens.is_cuda
            r.lens = lens.data_ptr()
        if quals is not None:
            assert quals.dtype == torch.uint8 and quals.is_cuda and quals.is_contiguous()
            r.quals = quals.data_ptr()
            r.qstride = quals.shape[1]
        if good_len is not None:
            assert good_len.dtype == torch.int16 and good_len.is_cuda
            r.good_len = good_len.data_ptr()
        if bc is not None:
            assert bc.dtype == torch.int32 and bc.is_cuda
            r.bc = bc.data_ptr()
        r.ign_bc_below = ign_bc_below
        r.read_index_base = read_index_base
        if group is not None:
            assert group.dtype == torch.int32 and group.is_cuda and group.is_contiguous()
            r.group = group.data_ptr()
        return self.count_graph_reads(r, params)

    # ---- streamed input: the job's reads arrive slab by slab (snk_dev_stream_begin / _append / _finish)
    def stream_begin(self, read_len: int, total_reads_ub: int, has_bc: bool = True, params: Params | None = None):
        self._stream_params = params or Params()
        p = self._stream_params.to_c()
        err = C.create_string_buffer(512)
        rc = self.lib.snk_dev_stream_begin(self._ctx, C.byref(p), int(read_len), int(total_reads_ub), 1 if has_bc else 0, self._stream(), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))

    def stream_append(self, rows: torch.Tensor, read_len: int, quals: torch.Tensor | None = None, bc: torch.Tensor | None = None,
                      lens: torch.Tensor | None = None, good_len: torch.Tensor | None = None, ign_bc_below: int = 0, read_index_base: int = 0):
        """One slab (asynchronous: its tensors must stay alive until the engine's stream has passed the call)."""
        assert rows.is_cuda and rows.dtype == torch.int32 and rows.is_contiguous()
        r = _lib.SnkDevReads()
        r.n_reads, r.rows, r.row_words, r.read_len = rows.shape[0], rows.data_ptr(), rows.shape[1], read_len
        if lens is not None:
            r.lens = lens.data_ptr()
        if quals is not None:
            assert quals.dtype == torch.uint8 and quals.is_cuda and quals.is_contiguous()
            r.quals, r.qstride = quals.data_ptr(), quals.shape[1]
        if good_len is not None:
            r.good_len = good_len.data_ptr()
        if bc is not None:
            r.bc = bc.data_ptr()
        r.ign_bc_below, r.read_index_base = ign_bc_below, read_index_base
        self.stream_append_reads(r)

    def stream_append_reads(self, r: "_lib.SnkDevReads"):
        err = C.create_string_buffer(512)
        rc = self.lib.snk_dev_stream_append(self._ctx, C.byref(r), self._stream(), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))

    def stream_finish(self) -> Result:
        raw = _lib.SnkDevResult()
        err = C.create_string_buffer(512)
        rc = self.lib.snk_dev_stream_finish(self._ctx, C.byref(raw), self._stream(), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        return Result(self, raw, self._stream_params.K)

    def count_graph_reads(self, r: "_lib.SnkDevReads", params: Params | None = None) -> Result:
        """The same for reads described by plain device pointers (e.g. the arrays of snk_dev_ingest_fasth)."""
        params = params or Params()
        p = params.to_c()
        raw = _lib.SnkDevResult()
        err = C.create_string_buffer(512)
        rc = self.lib.snk_dev_count_graph(self._ctx, C.byref(r), C.byref(p), C.byref(raw), self._stream(), err, 512)
        if rc != 0:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        return Result(self, raw, params.K)
