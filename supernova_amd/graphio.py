"""Host-side hand-off formats and the graph-from-unitigs step, through libsnk's C ABI.

  write_bv / read_bv   the `.bv` file tada writes and DF reads through MSPEDGES=
                       (lib/tada/src/debruijn.rs:895-929; BuildReadQGraph48.cc:1640-1642)
  hbv_from_unitigs     buildHBVFromEdges (lib/assembly/src/paths/long/HBVFromEdges.cc:244-296)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib

_CODE = np.zeros(256, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def unitigs_to_arrays(unitigs: list[str]):
    """ASCII unitigs -> (off u64[n+1], bases u8 codes)."""
    off = np.zeros(len(unitigs) + 1, dtype=np.uint64)
    if unitigs:
        off[1:] = np.cumsum([len(u) for u in unitigs], dtype=np.uint64)
    bases = _CODE[np.frombuffer("".join(unitigs).encode(), dtype=np.uint8)] if unitigs else np.zeros(0, np.uint8)
    return off, np.ascontiguousarray(bases)


def arrays_to_unitigs(off: np.ndarray, bases: np.ndarray) -> list[str]:
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[bases].tobytes().decode()
    return [asc[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]


def write_bv(path: str, off: np.ndarray, bases: np.ndarray) -> None:
    lib = _lib.load()
    off = np.ascontiguousarray(off, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    err = C.create_string_buffer(512)
    rc = lib.snk_write_bv(str(path).encode(), len(off) - 1, off.ctypes.data, bases.ctypes.data, err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))


def read_bv(path: str):
    lib = _lib.load()
    n = C.c_uint64(0)
    po = C.POINTER(C.c_uint64)()
    pb = C.POINTER(C.c_uint8)()
    err = C.create_string_buffer(512)
    rc = lib.snk_read_bv(str(path).encode(), C.byref(n), C.byref(po), C.byref(pb), err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    off = np.ctypeslib.as_array(po, shape=(n.value + 1,)).copy()
    tot = int(off[-1])
    bases = np.ctypeslib.as_array(pb, shape=(max(tot, 1),))[:tot].copy()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(po)
    libc.free(pb)
    return off, bases


def hbv_from_unitigs(K: int, off: np.ndarray, bases: np.ndarray) -> dict:
    """Unitigs in BVComp order -> HBV description (vertex ids per edge, fwd/rev translation)."""
    lib = _lib.load()
    off = np.ascontiguousarray(off, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    h = _lib.SnkHbv()
    err = C.create_string_buffer(512)
    rc = lib.snk_hbv_from_unitigs(K, len(off) - 1, off.ctypes.data, bases.ctypes.data, C.byref(h), err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    ne, nu = h.n_edges, len(off) - 1
    arr = lambda p, m, dt: (np.ctypeslib.as_array(p, shape=(m,)).astype(dt).copy() if m else np.zeros(0, dt))
    out = dict(n_vertices=h.n_vertices, n_edges=ne, v_left=arr(h.v_left, ne, np.int32), v_right=arr(h.v_right, ne, np.int32),
               src=arr(h.src_unitig, ne, np.int32), is_rc=arr(h.is_rc, ne, np.uint8), fwd=arr(h.fwd_xlat, nu, np.int32),
               rev=arr(h.rev_xlat, nu, np.int32))
    lib.snk_hbv_free(C.byref(h))
    return out


def write_hbv(path_hbv, path_inv, K: int, off: np.ndarray, bases: np.ndarray) -> np.ndarray:
    """Unitigs in BVComp order -> a.hbv (+ a.inv) as DF writes them; returns the involution."""
    lib = _lib.load()
    off = np.ascontiguousarray(off, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    h = _lib.SnkHbv()
    err = C.create_string_buffer(512)
    nu = len(off) - 1
    rc = lib.snk_hbv_from_unitigs(K, nu, off.ctypes.data, bases.ctypes.data, C.byref(h), err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    try:
        inv = np.zeros(max(h.n_edges, 1), dtype=np.int32)
        rc = lib.snk_hbv_involution(C.byref(h), nu, inv.ctypes.data, err, 512)
        if not rc:
            rc = lib.snk_write_hbv(str(path_hbv).encode(), str(path_inv).encode() if path_inv else None, K, nu, off.ctypes.data,
                                   bases.ctypes.data, C.byref(h), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        return inv[:h.n_edges].copy()
    finally:
        lib.snk_hbv_free(C.byref(h))


def hbv_text(unitigs: list[str], h: dict) -> str:
    """Same text layout as oracle/ref/ref_driver.cc's hbv.txt (golden fixtures)."""
    comp = str.maketrans("ACGT", "TGCA")
    lines = [f"N {h['n_vertices']} E {h['n_edges']} U {len(unitigs)}"]
    for e in range(h["n_edges"]):
        s = unitigs[h["src"][e]]
        if h["is_rc"][e]:
            s = s.translate(comp)[::-1]
        lines.append(f"E {e} {h['v_left'][e]} {h['v_right'][e]} {s}")
    for u in range(len(unitigs)):
        lines.append(f"X {u} {h['fwd'][u]} {h['rev'][u]}")
    return "\n".join(lines) + "\n"
