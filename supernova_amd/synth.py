"""Synthetic linked-read workloads through libsnk's counter-based generator."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib


def synth_params(n_reads: int, seed: int = 0x5EED0001, error_free: bool = False, **overrides) -> _lib.SnkSynthParams:
    sp = _lib.SnkSynthParams()
    _lib.load().snk_synth_default(C.byref(sp), n_reads, seed, 1 if error_free else 0)
    for k, v in overrides.items():
        if not hasattr(sp, k):
            raise AttributeError(k)
        setattr(sp, k, v)
    if "sub_ppm" in overrides or "read_len" in overrides:      # the error-count table follows the rate (and the read length)
        _lib.load().snk_synth_set_errors(C.byref(sp), int(sp.sub_ppm))
    return sp


def row_words_for(read_len: int) -> int:
    return (read_len + 15) // 16


def synth_host(sp: _lib.SnkSynthParams, first: int = 0, n: int | None = None, qstride: int | None = None):
    """Generate reads [first, first+n) on the host: (rows u32[n,row_words], quals u8[n,qstride], bc i32[n])."""
    n = sp.n_reads - first if n is None else n
    rw = row_words_for(sp.read_len)
    qs = qstride or sp.read_len
    rows = np.zeros((n, rw), dtype=np.uint32)
    quals = np.zeros((n, qs), dtype=np.uint8)
    bc = np.zeros(n, dtype=np.int32)
    _lib.check(_lib.load().snk_synth_host(C.byref(sp), first, n, rows.ctypes.data, rw, quals.ctypes.data, qs,
                                          bc.ctypes.data))
    return rows, quals, bc


def pack_rows(bases: np.ndarray) -> np.ndarray:
    """bases: u8[n, L] of codes 0..3 -> u32[n, ceil(L/16)] MSB-first packed rows."""
    n, L = bases.shape
    rw = row_words_for(L)
    pad = np.zeros((n, rw * 16), dtype=np.uint32)
    pad[:, :L] = bases & 3
    pad = pad.reshape(n, rw, 16)
    shifts = (30 - 2 * np.arange(16, dtype=np.uint32)).astype(np.uint32)
    return np.bitwise_or.reduce(pad << shifts, axis=2).astype(np.uint32)


def unpack_rows(rows: np.ndarray, read_len: int) -> np.ndarray:
    """u32[n, rw] packed rows -> u8[n, read_len] base codes."""
    n, rw = rows.shape
    shifts = (30 - 2 * np.arange(16, dtype=np.uint32)).astype(np.uint32)
    b = (rows[:, :, None] >> shifts) & 3
    return b.reshape(n, rw * 16)[:, :read_len].astype(np.uint8)


_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def codes_to_ascii(bases: np.ndarray) -> np.ndarray:
    return _ASCII[bases]


def ascii_to_codes(a: np.ndarray) -> np.ndarray:
    """ASCII -> 2-bit codes; every non-ACGT character maps to A (kmer/mod.rs:311-319)."""
    lut = np.zeros(256, dtype=np.uint8)
    lut[ord("C")] = 1
    lut[ord("G")] = 2
    lut[ord("T")] = 3
    return lut[a]
