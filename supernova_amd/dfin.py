"""The ASSEMBLER_DF stage inputs (reads.fastb / reads.qualp / reads.bci) decoded on the device (snk_dfin.hip, include/snk.h):
`DfFiles.ingest` leaves a range of reads resident in HBM, `DfFiles.count_graph` streams them slab by slab into a count+graph job.
The reference: bases.ReadAll + VirtualMasterVec<PQVec> + the barcode index expansion (lib/assembly/src/10X/DF.cc:265-272,345,464-469,
595-597).  `write_df` / `write_synth_df` make test and bench inputs."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import lib as _lib
from .ingest import DeviceReads


def _err():
    return C.create_string_buffer(512)


def _check(rc, err):
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))


class DfFiles:
    def __init__(self, head, with_bci: bool = True):
        head = str(head)
        if head.endswith(".fastb"):
            head = head[: -len(".fastb")]
        self.lib = _lib.load()
        self.head = head
        self._h = C.c_void_p()
        info = _lib.SnkDfInfo()
        err = _err()
        bci = (head + ".bci").encode() if with_bci else None
        _check(self.lib.snk_df_open((head + ".fastb").encode(), (head + ".qualp").encode(), bci, C.byref(self._h), C.byref(info), err, 512), err)
        self.n_reads, self.n_barcodes = int(info.n_reads), int(info.n_barcodes)
        self.file_bytes = int(info.fastb_bytes) + int(info.qualp_bytes) + int(info.bci_bytes)

    def close(self):
        if self._h:
            self.lib.snk_df_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def max_len(self, engine, first: int = 0, n: int | None = None) -> int:
        n = self.n_reads - first if n is None else n
        out, err = C.c_uint32(0), _err()
        _check(self.lib.snk_df_max_len(engine._ctx, self._h, first, n, C.byref(out), err, 512), err)
        return int(out.value)

    def ingest(self, engine, first: int = 0, n: int | None = None, read_len: int = 0, threads: int = 0, slab_reads: int = 0) -> DeviceReads:
        n = self.n_reads - first if n is None else n
        raw, err = _lib.SnkDevIngest(), _err()
        _check(self.lib.snk_dev_ingest_df(engine._ctx, self._h, first, n, read_len, threads, slab_reads, C.byref(raw), err, 512), err)
        return DeviceReads(self.lib, raw)

    def ingest_trimmed(self, engine, K: int = 48, min_qual: int = 7, first: int = 0, n: int | None = None, read_len: int = 0, threads: int = 0,
                       slab_reads: int = 0) -> DeviceReads:
        """The compact form (snk_dev_ingest_df_trimmed): packed rows, good lengths, barcode ids -- no quality rows, no lengths."""
        n = self.n_reads - first if n is None else n
        raw, err = _lib.SnkDevIngest(), _err()
        _check(self.lib.snk_dev_ingest_df_trimmed(engine._ctx, self._h, first, n, read_len, threads, slab_reads, K, min_qual, C.byref(raw), err, 512), err)
        return DeviceReads(self.lib, raw)

    def count_graph(self, engine, params=None, first: int = 0, n: int | None = None, read_len: int = 0, threads: int = 0, slab_reads: int = 0,
                    ign_bc_below: int = 0):
        """-> (Result, stats); bit-identical to a resident call on the same reads.  stats["mode"]: "compact" (default: rows + good lengths +
        barcode ids stay, the resident step with its pilot / second partition / kernel choice runs on them) or "streamed"
        (engine.set_option("df_stream", 2): the slabs go straight into a streamed job, the reads are never resident in any form)."""
        from .engine import Params, Result
        params = params or Params()
        n = self.n_reads - first if n is None else n
        raw, res, p, err = _lib.SnkDevIngest(), _lib.SnkDevResult(), params.to_c(), _err()
        _check(self.lib.snk_dev_ingest_df_count_graph(engine._ctx, self._h, first, n, read_len, threads, slab_reads, C.byref(p), ign_bc_below,
                                                      C.byref(res), C.byref(raw), err, 512), err)
        stats = dict(n_reads=int(raw.n_reads), file_bytes=int(raw.text_bytes), seconds=float(raw.seconds), io_wait_seconds=float(raw.decode_wait_seconds),
                     n_slabs=int(raw.n_batches), max_len=int(raw.max_len), setup_seconds=float(raw.setup_seconds),
                     mode="streamed" if int(raw.n_files) == 3 else "compact")
        return Result(engine, res, params.K), stats


def write_df(head, rows: np.ndarray, quals: np.ndarray, bc: np.ndarray | None = None, lens: np.ndarray | None = None, read_len: int | None = None,
             threads: int = 0, adversarial: int = 0) -> None:
    """<head>.fastb / .qualp / .bci from host arrays (rows u32[n, row_words] MSB-first packed, quals u8[n, qstride] raw phred, bc i32[n] ordered)."""
    lib = _lib.load()
    rows = np.ascontiguousarray(rows, dtype=np.uint32)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    n = rows.shape[0]
    read_len = int(read_len if read_len is not None else quals.shape[1])
    lp = np.ascontiguousarray(lens, dtype=np.uint16) if lens is not None else None
    bp = np.ascontiguousarray(bc, dtype=np.int32) if bc is not None else None
    err = _err()
    _check(lib.snk_write_df(str(head).encode(), n, rows.ctypes.data, rows.shape[1], lp.ctypes.data if lp is not None else None, read_len, quals.ctypes.data,
                            quals.shape[1], bp.ctypes.data if bp is not None else None, threads, adversarial, err, 512), err)


def write_synth_df(head, sp: _lib.SnkSynthParams, first: int = 0, n: int | None = None, qual_jitter: int = 0, threads: int = 0) -> None:
    lib = _lib.load()
    n = int(sp.n_reads) - first if n is None else n
    Path(str(head)).parent.mkdir(parents=True, exist_ok=True)
    err = _err()
    _check(lib.snk_synth_df_write(str(head).encode(), C.byref(sp), first, n, qual_jitter, threads, err, 512), err)
