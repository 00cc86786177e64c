"""Readers for ASSEMBLER_DF's stage inputs (reads.fastb / reads.qualp / reads.bci), through libsnk's C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib


def _free(p):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(p)


def read_fastb(path):
    """-> (rows u32[n, row_words], lens u16[n], max_len)"""
    lib = _lib.load()
    n, mx = C.c_uint64(0), C.c_uint32(0)
    pl, pr = C.POINTER(C.c_uint16)(), C.POINTER(C.c_uint32)()
    err = C.create_string_buffer(512)
    rc = lib.snk_read_fastb(str(path).encode(), C.byref(n), C.byref(mx), C.byref(pl), C.byref(pr), err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    rw = max(1, (mx.value + 15) // 16)
    lens = np.ctypeslib.as_array(pl, shape=(max(n.value, 1),))[: n.value].copy()
    rows = np.ctypeslib.as_array(pr, shape=(max(n.value * rw, 1),))[: n.value * rw].copy().reshape(n.value, rw)
    _free(pl)
    _free(pr)
    return rows, lens, int(mx.value)


def read_qualp(path, n_reads: int, qstride: int) -> np.ndarray:
    lib = _lib.load()
    q = np.zeros((n_reads, qstride), dtype=np.uint8)
    err = C.create_string_buffer(512)
    rc = lib.snk_read_qualp(str(path).encode(), n_reads, qstride, q.ctypes.data, err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    return q


def read_bci(path, n_reads: int):
    """-> (bc i32[n_reads] barcode ordinal per read, 0 = unbarcoded block; n_barcodes)"""
    lib = _lib.load()
    bc = np.zeros(n_reads, dtype=np.int32)
    nb = C.c_uint64(0)
    err = C.create_string_buffer(512)
    rc = lib.snk_read_bci(str(path).encode(), n_reads, bc.ctypes.data, C.byref(nb), err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    return bc, int(nb.value)
