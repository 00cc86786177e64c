"""f3: FASTH files -> reads resident in HBM at rate (snk_dev_ingest_fasth, include/snk.h).

The reference reads FASTH on one thread per two files and looks every barcode up in a hash map (lib/tada/src/cmd_msp.rs:55-69,
multifastq.rs:69-127, utils.rs:101-164); here a pool of decode threads fills page-locked batches, the uploads, the 2-bit
pack and the barcode ids (one kernel per batch) overlap the decode.  `write_synth_fasth` makes test / bench inputs from the
synthetic read model.
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

from . import lib as _lib


def write_synth_fasth(directory, sp: _lib.SnkSynthParams, n_files: int, pairs_per_file: int, level: int = 1, workers: int = 32):
    """n_files gzip FASTH files with consecutive pairs of the synthetic data set sp -> (paths, inflated bytes)."""
    lib = _lib.load()
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    paths = [str(directory / f"chunk{i:04d}.fasth.gz") for i in range(n_files)]

    def mk(i):
        err = C.create_string_buffer(512)
        tb = C.c_uint64(0)
        rc = lib.snk_synth_fasth_write(paths[i].encode(), C.byref(sp), i * pairs_per_file, pairs_per_file, level, C.byref(tb), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        return int(tb.value)

    with ThreadPoolExecutor(max(1, min(workers, n_files))) as ex:      # (the C call releases the GIL)
        text = sum(ex.map(mk, range(n_files)))
    return paths, text


def synth_whitelist(n_ids: int) -> bytes:
    """Whitelist whose line i (0-based) is the sequence of synthetic barcode id i + 1."""
    lib = _lib.load()
    buf = C.create_string_buffer(16)
    out = bytearray()
    for i in range(1, n_ids + 1):
        lib.snk_synth_bc_seq(i, buf)
        out += buf.raw + b"\n"
    return bytes(out)


class DeviceReads:
    """Reads that snk_dev_ingest_fasth left in HBM (plain device allocations, released by close())."""

    def __init__(self, lib, raw: _lib.SnkDevIngest):
        self.lib, self.raw = lib, raw
        self.n_reads, self.read_len = int(raw.n_reads), int(raw.read_len)
        self.stats = dict(n_reads=self.n_reads, text_bytes=int(raw.text_bytes), compressed_bytes=int(raw.compressed_bytes),
                          seconds=float(raw.seconds), decode_wait_seconds=float(raw.decode_wait_seconds), n_files=int(raw.n_files),
                          n_batches=int(raw.n_batches), max_len=int(raw.max_len), setup_seconds=float(raw.setup_seconds))

    def dev_reads(self, with_bc: bool = True) -> _lib.SnkDevReads:
        r = _lib.SnkDevReads()
        r.n_reads, r.rows, r.row_words, r.read_len = self.n_reads, self.raw.rows, self.raw.row_words, self.read_len
        r.lens, r.quals, r.qstride = self.raw.lens, self.raw.quals, self.raw.qstride
        if self.raw.good_len:                       # the compact form of snk_dev_ingest_df_trimmed: no quality rows, the trim is done
            r.good_len = self.raw.good_len
        if with_bc and self.raw.bc:
            r.bc = self.raw.bc
        return r

    def close(self):
        if self.raw is not None:
            self.lib.snk_dev_ingest_free(C.byref(self.raw))
            self.raw = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ingest_fasth(engine, paths, read_len: int, whitelist: bytes | None = None, threads: int = 0, batch_pairs: int = 0) -> DeviceReads:
    lib = engine.lib
    err = C.create_string_buffer(512)
    ix = C.c_void_p()
    if whitelist is not None:
        rc = lib.snk_bc_index_create(engine._ctx, whitelist, len(whitelist), C.byref(ix), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    try:
        arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
        raw = _lib.SnkDevIngest()
        rc = lib.snk_dev_ingest_fasth(engine._ctx, arr, len(paths), read_len, ix, threads, batch_pairs, C.byref(raw), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    finally:
        if ix:
            lib.snk_bc_index_destroy(ix)
    return DeviceReads(lib, raw)


def ingest_count_graph(engine, paths, read_len: int, whitelist: bytes | None = None, params=None, threads: int = 0, batch_pairs: int = 0,
                       total_reads_hint: int = 0):
    """FASTH files -> (Result, stats) with the reads never resident as a whole (snk_dev_ingest_count_graph): every decoded batch is
    partitioned into the job's minimiser buckets while the next ones are being inflated."""
    from .engine import Params, Result
    lib = engine.lib
    params = params or Params()
    err = C.create_string_buffer(512)
    ix = C.c_void_p()
    if whitelist is not None:
        rc = lib.snk_bc_index_create(engine._ctx, whitelist, len(whitelist), C.byref(ix), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    try:
        arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
        raw, res, p = _lib.SnkDevIngest(), _lib.SnkDevResult(), params.to_c()
        rc = lib.snk_dev_ingest_count_graph(engine._ctx, arr, len(paths), read_len, ix, threads, batch_pairs, int(total_reads_hint), C.byref(p), C.byref(res),
                                            C.byref(raw), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    finally:
        if ix:
            lib.snk_bc_index_destroy(ix)
    stats = dict(n_reads=int(raw.n_reads), text_bytes=int(raw.text_bytes), compressed_bytes=int(raw.compressed_bytes), seconds=float(raw.seconds),
                 decode_wait_seconds=float(raw.decode_wait_seconds), n_files=int(raw.n_files), n_batches=int(raw.n_batches), max_len=int(raw.max_len),
                 setup_seconds=float(raw.setup_seconds))
    return Result(engine, res, params.K), stats
