"""ctypes wrapper of oracle/libsnkoracle.so -- TEST INFRASTRUCTURE (the checker, never the product)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
LIB = ORACLE_DIR / "libsnkoracle.so"


class Table(C.Structure):
    _fields_ = [("n", C.c_uint64), ("key", C.POINTER(C.c_uint32)), ("count", C.POINTER(C.c_uint32)),
                ("ctx_raw", C.POINTER(C.c_uint8)), ("ctx", C.POINTER(C.c_uint8))]


class Unitigs(C.Structure):
    _fields_ = [("n", C.c_uint64), ("off", C.POINTER(C.c_uint64)), ("bases", C.POINTER(C.c_uint8))]


class Hbv(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_edges", C.c_int32), ("v_left", C.POINTER(C.c_int32)),
                ("v_right", C.POINTER(C.c_int32)), ("src_unitig", C.POINTER(C.c_int32)),
                ("is_rc", C.POINTER(C.c_uint8)), ("fwd_xlat", C.POINTER(C.c_int32)),
                ("rev_xlat", C.POINTER(C.c_int32))]


class Slice(C.Structure):
    _fields_ = [("value", C.c_uint32), ("min_pos", C.c_uint32), ("start", C.c_uint32), ("len", C.c_uint32)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not LIB.exists() or LIB.stat().st_mtime < (ORACLE_DIR / "snk_oracle.c").stat().st_mtime:
            subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, capture_output=True)
        lib = C.CDLL(str(LIB))
        lib.sno_good_len.restype = C.c_uint32
        lib.sno_good_len.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.sno_msp_scan.restype = C.c_int
        lib.sno_msp_scan.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Slice), C.c_int]
        lib.sno_count.restype = C.c_int
        lib.sno_count.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                  C.c_uint32, C.c_int64, C.POINTER(Table), C.POINTER(C.c_uint64)]
        lib.sno_unitigs_build.restype = C.c_int
        lib.sno_unitigs_build.argtypes = [C.POINTER(Table), C.c_uint32, C.POINTER(Unitigs)]
        lib.sno_hbv_build.restype = C.c_int
        lib.sno_hbv_build.argtypes = [C.POINTER(Unitigs), C.c_uint32, C.POINTER(Hbv)]
        lib.sno_write_bv.restype = C.c_int
        lib.sno_write_bv.argtypes = [C.c_char_p, C.POINTER(Unitigs)]
        lib.sno_read_bv.restype = C.c_int
        lib.sno_read_bv.argtypes = [C.c_char_p, C.POINTER(Unitigs)]
        lib.sno_path_reads.restype = C.c_int
        lib.sno_path_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(Unitigs),
                                       C.POINTER(Hbv), C.c_void_p, C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_uint64)]
        lib.sno_free.restype = None
        lib.sno_free.argtypes = [C.c_void_p]
        for f in ("sno_table_free", "sno_unitigs_free", "sno_hbv_free"):
            getattr(lib, f).restype = None
        _lib = lib
    return _lib


def good_lens(quals: np.ndarray, lens, K=48, min_qual=7) -> np.ndarray:
    lib = load()
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    n, stride = quals.shape
    lens = np.broadcast_to(np.asarray(lens), (n,))
    out = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        out[i] = lib.sno_good_len(quals[i].ctypes.data, int(lens[i]), K, min_qual)
    return out


def msp_scan(k, p, seq_codes: np.ndarray, perm=None):
    lib = load()
    seq = np.ascontiguousarray(seq_codes, dtype=np.uint8)
    cap = len(seq) + 4
    buf = (Slice * cap)()
    permp = None
    if perm is not None:
        perm = np.ascontiguousarray(perm, dtype=np.uint32)
        permp = perm.ctypes.data
    ns = lib.sno_msp_scan(k, p, seq.ctypes.data, len(seq), permp, buf, cap)
    if ns < 0:
        raise RuntimeError("sno_msp_scan failed")
    return [(buf[i].value, buf[i].min_pos, buf[i].start, buf[i].len) for i in range(ns)]


def path_reads(codes: np.ndarray, quals: np.ndarray, lens, unitigs: list[str], K=48):
    """f1: read paths of untrimmed reads on the graph of the given (BVComp-ordered) unitigs: (offset i32[n], n_edges i32[n],
    edges i32[sum]) with HBV edge ids -- the restatement of pathReads / HBVPather::algorithmTwo / ExtendReadPath."""
    lib = load()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    n, stride = codes.shape
    assert quals.shape == codes.shape
    lens = np.ascontiguousarray(np.broadcast_to(np.asarray(lens), (n,)), dtype=np.uint32)
    lut = np.zeros(256, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    off = np.zeros(len(unitigs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(u) for u in unitigs])
    bases = lut[np.frombuffer("".join(unitigs).encode(), dtype=np.uint8)] if unitigs else np.zeros(1, np.uint8)
    bases = np.ascontiguousarray(bases)
    u = Unitigs(len(unitigs), off.ctypes.data_as(C.POINTER(C.c_uint64)), bases.ctypes.data_as(C.POINTER(C.c_uint8)))
    h = Hbv()
    if lib.sno_hbv_build(C.byref(u), K, C.byref(h)) != 0:
        raise RuntimeError("sno_hbv_build failed")
    o_off = np.zeros(n, dtype=np.int32)
    o_n = np.zeros(n, dtype=np.int32)
    pe = C.POINTER(C.c_int32)()
    tot = C.c_uint64(0)
    rc = lib.sno_path_reads(codes.ctypes.data, quals.ctypes.data, stride, lens.ctypes.data, n, K, C.byref(u), C.byref(h),
                            o_off.ctypes.data, o_n.ctypes.data, C.byref(pe), C.byref(tot))
    lib.sno_hbv_free(C.byref(h))
    if rc != 0:
        raise RuntimeError(f"sno_path_reads failed {rc}")
    edges = np.ctypeslib.as_array(pe, shape=(max(tot.value, 1),))[:tot.value].copy()
    lib.sno_free(pe)
    return o_off, o_n, edges


class OracleResult:
    """Runs count (+ optional unitigs/HBV) and copies everything into numpy arrays."""

    def __init__(self, bases_codes, good_len, bc, K=48, min_freq=3, min_bc=2, ign_bc_below=0, graph=True, hbv=True):
        lib = load()
        bases = np.ascontiguousarray(bases_codes, dtype=np.uint8)
        n, stride = bases.shape
        gl = np.ascontiguousarray(good_len, dtype=np.uint32)
        bcp = None
        if bc is not None:
            bcarr = np.ascontiguousarray(bc, dtype=np.int32)
            bcp = bcarr.ctypes.data
        t = Table()
        ninst = C.c_uint64(0)
        rc = lib.sno_count(bases.ctypes.data, stride, gl.ctypes.data, bcp, n, K, min_freq, min_bc, ign_bc_below,
                           C.byref(t), C.byref(ninst))
        if rc != 0:
            raise RuntimeError(f"sno_count failed {rc}")
        self.K = K
        self.n_instances = ninst.value
        nk = t.n
        self.keys = np.ctypeslib.as_array(t.key, shape=(nk, 4)).copy() if nk else np.zeros((0, 4), np.uint32)
        self.counts = np.ctypeslib.as_array(t.count, shape=(nk,)).copy() if nk else np.zeros(0, np.uint32)
        self.ctx_raw = np.ctypeslib.as_array(t.ctx_raw, shape=(nk,)).copy() if nk else np.zeros(0, np.uint8)
        self.ctx = np.ctypeslib.as_array(t.ctx, shape=(nk,)).copy() if nk else np.zeros(0, np.uint8)
        self.unitigs = None
        self.hbv = None
        if graph:
            u = Unitigs()
            rc = lib.sno_unitigs_build(C.byref(t), K, C.byref(u))
            if rc != 0:
                raise RuntimeError(f"sno_unitigs_build failed {rc}")
            off = np.ctypeslib.as_array(u.off, shape=(u.n + 1,)).copy()
            tot = int(off[-1])
            b = np.ctypeslib.as_array(u.bases, shape=(max(tot, 1),))[:tot].copy()
            self.unitig_off = off
            self.unitig_bases = b
            asc = np.frombuffer(b"ACGT", dtype=np.uint8)[b].tobytes().decode()
            self.unitigs = [asc[int(off[i]):int(off[i + 1])] for i in range(u.n)]
            if hbv:
                h = Hbv()
                rc = lib.sno_hbv_build(C.byref(u), K, C.byref(h))
                if rc != 0:
                    raise RuntimeError(f"sno_hbv_build failed {rc}")
                ne, nu = h.n_edges, u.n
                arr = lambda p, m, dt: (np.ctypeslib.as_array(p, shape=(m,)).astype(dt).copy() if m else np.zeros(0, dt))
                self.hbv = dict(n_vertices=h.n_vertices, n_edges=ne, v_left=arr(h.v_left, ne, np.int32),
                                v_right=arr(h.v_right, ne, np.int32), src=arr(h.src_unitig, ne, np.int32),
                                is_rc=arr(h.is_rc, ne, np.uint8), fwd=arr(h.fwd_xlat, nu, np.int32),
                                rev=arr(h.rev_xlat, nu, np.int32))
                lib.sno_hbv_free(C.byref(h))
            lib.sno_unitigs_free(C.byref(u))
        lib.sno_table_free(C.byref(t))

    def hbv_text(self) -> str:
        """Same text layout as oracle/ref/ref_driver.cc's hbv.txt."""
        comp = str.maketrans("ACGT", "TGCA")
        h = self.hbv
        lines = [f"N {h['n_vertices']} E {h['n_edges']} U {len(self.unitigs)}"]
        for e in range(h["n_edges"]):
            s = self.unitigs[h["src"][e]]
            if h["is_rc"][e]:
                s = s.translate(comp)[::-1]
            lines.append(f"E {e} {h['v_left'][e]} {h['v_right'][e]} {s}")
        for u in range(len(self.unitigs)):
            lines.append(f"X {u} {h['fwd'][u]} {h['rev'][u]}")
        return "\n".join(lines) + "\n"


def mark_dups(codes: np.ndarray, quals: np.ndarray, lens, path_off, path_n, path_edges, bc=None):
    """f4: the restatement of MarkDups (10X/SecretOps.cc:413-593) over read paths in the dumped form (offset, edge count, edges
    concatenated).  Returns (dup u8[n/2], art u8[n/2], interdup_rate, n_dups, n_interdups)."""
    lib = load()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    n, stride = codes.shape
    lens = np.ascontiguousarray(np.broadcast_to(np.asarray(lens), (n,)), dtype=np.uint32)
    path_n = np.asarray(path_n, dtype=np.int64)
    start = np.zeros(n + 1, dtype=np.int64)
    start[1:] = np.cumsum(path_n)
    pe = np.asarray(path_edges, dtype=np.int32)
    first = np.full(n, -1, dtype=np.int32)
    has = path_n > 0
    first[has] = pe[start[:-1][has]]
    off = np.ascontiguousarray(path_off, dtype=np.int32)
    bcp = None
    if bc is not None:
        bca = np.ascontiguousarray(bc, dtype=np.int32)
        bcp = bca.ctypes.data
    dup = np.zeros(n // 2, dtype=np.uint8)
    art = np.zeros(n // 2, dtype=np.uint8)
    rate = C.c_double(0)
    nd, ni = C.c_uint64(0), C.c_uint64(0)
    lib.sno_mark_dups.restype = C.c_int
    lib.sno_mark_dups.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    rc = lib.sno_mark_dups(codes.ctypes.data, quals.ctypes.data, stride, lens.ctypes.data, n, first.ctypes.data, off.ctypes.data, bcp,
                           dup.ctypes.data, art.ctypes.data, C.byref(rate), C.byref(nd), C.byref(ni))
    if rc != 0:
        raise RuntimeError(f"sno_mark_dups failed {rc}")
    return dup, art, float(rate.value), int(nd.value), int(ni.value)


def unitig_barcodes(codes: np.ndarray, lens, bc, unitigs: list[str], K=48):
    """The rest of f4, restated in plain Python (PARITY UNPINNED: the reference side is Rust -- tada's MAIN_ASM_SN,
    lib/tada/src/cmd_main_asm.rs:91-151 with barcodes_for_sedge, debruijn.rs:115-131 -- which cannot be built here): per unitig
    the sorted distinct barcodes > 0 of the reads that have at least one k-mer of the unitig (either strand).  unitigs: strings,
    the numbering the result refers to.  Returns a list of sorted lists."""
    comp = str.maketrans("ACGT", "TGCA")
    where = {}
    for u, sq in enumerate(unitigs):
        for i in range(len(sq) - K + 1):
            k = sq[i:i + K]
            r = k.translate(comp)[::-1]
            where[k if k <= r else r] = u
    n = codes.shape[0]
    lens = np.broadcast_to(np.asarray(lens), (n,))
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    out = [set() for _ in unitigs]
    for r in range(n):
        b = int(bc[r])
        if b <= 0:
            continue
        sq = asc[r, :int(lens[r])].tobytes().decode()
        for i in range(len(sq) - K + 1):
            k = sq[i:i + K]
            rk = k.translate(comp)[::-1]
            u = where.get(k if k <= rk else rk)
            if u is not None:
                out[u].add(b)
    return [sorted(x) for x in out]
