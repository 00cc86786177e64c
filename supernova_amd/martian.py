"""Martian stage adapter: the `_ASM_SN` operator surface (lib/tada/mro/_asm_sn.mro:7-33) on one MI355X.

It speaks the exec-stage protocol of the reference's Rust adapter (lib/tada/external/martian/src/lib.rs):
argv `<exe> martian <stage> <split|main|join> <metadata_path> <files_path> <run_file>` (:142-159), reads
`_args`/`_outs`/`_chunk_defs`/`_chunk_outs`, writes `_stage_defs`/`_outs`/`_complete`, errors go to `_errors`
(:568-602), every file it writes is journalled as `<run_file>.<name>` (:181-200), `_log` lines carry the same
timestamp format (:135-141, 244-251).

Stage ASM_SN_GPU replaces MSP + SHARD_ASM + MAIN_ASM_SN:
  FASTH parse          lib/tada/src/multifastq.rs:72-126     (9-line records, barcode "SEQ-gemgroup[,raw]")
  barcode ids          lib/tada/src/utils.rs:101-164          (whitelist index + 1 + (gem_group-1)*N, 0 = not whitelisted)
  count + graph        libsnk (include/snk.h) -- semantics of path B (SURVEY.md App. A.9 lists where tada differs:
                       a read trimmed to exactly K bases contributes no k-mer here)
  asm_graph.bv         lib/tada/src/debruijn.rs:895-929
"""
from __future__ import annotations

import datetime
import gzip
import json
import os
import sys
import traceback
from pathlib import Path

import numpy as np

METADATA_PREFIX = "_"


# ------------------------------------------------------------------------------------------------ ingest
class BcIndexer:
    """lib/tada/src/utils.rs:101-164."""

    def __init__(self, lines):
        self.bc_map = {}
        i = 0
        for l in lines:
            l = l[:-1] if l.endswith("\n") else l
            self.bc_map[l[:-1] if l.endswith("\r") else l] = i       # BufRead::lines strips "\n" or "\r\n"
            i += 1
        self.num_bcs = i

    @classmethod
    def from_file(cls, path):
        with open(path) as f:
            return cls(f)

    def get_bc_id(self, bc: str):
        if "-" in bc:
            seq, gg = bc.split("-")[:2]
            idx = self.bc_map.get(seq)
            # u8::from_str: decimal digits with an optional leading '+', 0..255, anything else panics ("invalid gem group string")
            if not (gg[1:] if gg[:1] == "+" else gg).isdigit() or not gg.isascii() or int(gg) > 255:
                raise ValueError("invalid gem group string")
            g = int(gg)
        else:
            idx = self.bc_map.get(bc)
            g = 1
        if idx is None:
            return None
        if g == 0 or (g - 1) * self.num_bcs >= 1 << 32:
            raise OverflowError("too many gem groups - BC id overflowed")      # (g - 1) underflows / checked_mul fails, utils.rs:157
        return (g - 1) * self.num_bcs + idx + 1


class DeviceBcIndexer:
    """The same mapping computed on the MI355X (snk_bc_index_create / snk_dev_bc_ids, include/snk.h): the whitelist lives
    in HBM, the barcode fields of all reads are looked up by one kernel."""
    FIELD = 64

    def __init__(self, whitelist_bytes: bytes, device: int = 0):
        import ctypes as C
        from . import lib as _lib
        self._C, self._lib_mod = C, _lib
        self.lib = _lib.load()
        self.device = device
        self._ctx = C.c_void_p()
        self._ix = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self.lib.snk_ctx_create(device, C.byref(self._ctx), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        rc = self.lib.snk_bc_index_create(self._ctx, whitelist_bytes, len(whitelist_bytes), C.byref(self._ix), err, 512)
        if rc:
            self.lib.snk_ctx_destroy(self._ctx)
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        self.num_bcs = int(self.lib.snk_bc_index_lines(self._ix))

    @classmethod
    def from_file(cls, path, device: int = 0):
        with open(path, "rb") as f:
            return cls(f.read(), device)

    def ids_of_fields(self, fields: np.ndarray) -> np.ndarray:
        """fields: uint8 [n, stride], one zero-padded barcode field ("SEQ[-gg][,raw]") per row -> int32 ids (0 = none)."""
        import torch
        C = self._C
        fields = np.ascontiguousarray(fields, dtype=np.uint8)
        n, stride = fields.shape
        dev = torch.device("cuda", self.device)
        d_f = torch.from_numpy(fields).to(dev)
        d_ids = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        err = C.create_string_buffer(512)
        rc = self.lib.snk_dev_bc_ids(self._ctx, self._ix, d_f.data_ptr(), stride, n, d_ids.data_ptr(), None, err, 512)
        if rc:
            raise self._lib_mod.SnkError(rc, err.value.decode(errors="replace"))
        return d_ids[:n].cpu().numpy()

    def close(self):
        if self._ix:
            self.lib.snk_bc_index_destroy(self._ix)
            self._ix = None
        if self._ctx:
            self.lib.snk_ctx_destroy(self._ctx)
            self._ctx = None


def read_fasth_native(paths):
    """The C++ reader of libsnk (snk_read_fasth, zlib): -> (ascii u8[n,L], quals u8[n,L], lens u16[n], fields u8[n/2,64])."""
    import ctypes as C
    from . import lib as _lib
    lib = _lib.load()
    STRIDE = 256
    parts = []
    for p in paths:
        n, mx = C.c_uint64(0), C.c_uint32(0)
        pa, pq, pf = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint8)()
        pl = C.POINTER(C.c_uint16)()
        err = C.create_string_buffer(512)
        rc = lib.snk_read_fasth(str(p).encode(), STRIDE, C.byref(n), C.byref(mx), C.byref(pa), C.byref(pq), C.byref(pl), C.byref(pf), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        nr = int(n.value)
        take = lambda ptr, shape, dt: (np.ctypeslib.as_array(ptr, shape=shape).astype(dt).copy() if nr else np.zeros(shape, dt))
        parts.append((take(pa, (nr, STRIDE), np.uint8), take(pq, (nr, STRIDE), np.uint8), take(pl, (nr,), np.uint16),
                      take(pf, (nr // 2, 64), np.uint8), int(mx.value)))
        for ptr in (pa, pq, pl, pf):
            lib.snk_host_free(ptr)
    L = max([x[4] for x in parts] + [1])
    cat = lambda i: np.concatenate([x[i] for x in parts]) if parts else np.zeros((0,), np.uint8)
    return cat(0)[:, :L].copy(), cat(1)[:, :L].copy(), cat(2), cat(3)


def read_fasth_stream(paths, threads=0, batch_pairs=0, stride=256):
    """The streaming reader of libsnk (snk_fasth_open / _next: `threads` files decoded concurrently, batches in any order) put back
    into file-major order: -> (ascii u8[n,L], quals u8[n,L], lens u16[n], fields u8[n/2,64], stats)."""
    import ctypes as C
    from . import lib as _lib
    lib = _lib.load()
    arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = lib.snk_fasth_open(arr, len(paths), stride, batch_pairs, threads, 0, C.byref(h), err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    got, text, nb = [], 0, 0
    try:
        while True:
            b = _lib.SnkFasthBatch()
            rc = lib.snk_fasth_next(h, C.byref(b), err, 512)
            if rc:
                raise _lib.SnkError(rc, err.value.decode(errors="replace"))
            if b.n_pairs == 0:
                break
            npair = int(b.n_pairs)
            view = lambda ptr, shape, dt: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(np.prod(shape)) * np.dtype(dt).itemsize,)).view(dt).reshape(shape).copy()
            got.append((int(b.file), int(b.first_pair), view(b.ascii, (2 * npair, stride), np.uint8), view(b.quals, (2 * npair, stride), np.uint8),
                        view(b.lens, (2 * npair,), np.uint16), view(b.bc_fields, (npair, 64), np.uint8)))
            text += int(b.text_bytes)
            nb += 1
            lib.snk_fasth_release(h, C.byref(b))
        pairs = [int(lib.snk_fasth_file_pairs(h, i)) for i in range(len(paths))]
    finally:
        lib.snk_fasth_close(h)
    got.sort(key=lambda t: (t[0], t[1]))
    for fi in range(len(paths)):          # the batches of a file tile it exactly
        at = 0
        for t in got:
            if t[0] == fi:
                assert t[1] == at
                at += t[5].shape[0]
        assert at == pairs[fi]
    cat = lambda i, shape, dt: np.concatenate([t[i] for t in got]) if got else np.zeros(shape, dt)
    lens = cat(4, (0,), np.uint16)
    L = max(int(lens.max()) if lens.size else 1, 1)
    return (cat(2, (0, stride), np.uint8)[:, :L].copy(), cat(3, (0, stride), np.uint8)[:, :L].copy(), lens, cat(5, (0, 64), np.uint8),
            dict(text_bytes=text, batches=nb, file_pairs=pairs))


def read_fasth(paths, indexer):
    """FASTH records -> (ascii u8[n,L], quals u8[n,L] raw phred, lens u16[n], bc i32[n]); R1 = read 2q, R2 = 2q+1
    (cmd_msp.rs:160-181).  With a device indexer the files are parsed by the library's C++ reader and the barcode ids
    come from one kernel; the pure-Python parser below is the restatement the tests compare it with."""
    if hasattr(indexer, "ids_of_fields"):
        asc, qa, lens, fields = read_fasth_native(paths)
        ids = indexer.ids_of_fields(fields) if len(fields) else np.zeros(0, np.int32)
        return asc, qa, lens, np.repeat(np.asarray(ids, dtype=np.int32), 2)
    seqs, quals, bcs = [], [], []
    for p in paths:
        with gzip.open(p, "rt") as f:
            while True:
                head = f.readline()
                if not head:
                    break
                r1, q1, r2, q2 = (f.readline().rstrip("\n") for _ in range(4))
                bc = f.readline().rstrip("\n")
                for _ in range(3):
                    f.readline()
                seq = bc.split(",")[0] if "," in bc else bc
                seqs += [r1, r2]
                quals += [q1, q2]
                bcs.append(seq)
    ids = np.array([indexer.get_bc_id(sq) or 0 for sq in bcs], dtype=np.int32)
    bcs = np.repeat(np.asarray(ids, dtype=np.int32), 2)
    n = len(seqs)
    L = max((len(s) for s in seqs), default=1)
    asc = np.full((n, L), ord("A"), dtype=np.uint8)
    qa = np.zeros((n, L), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    for i, (s, q) in enumerate(zip(seqs, quals)):
        lens[i] = len(s)
        asc[i, :len(s)] = np.frombuffer(s.encode(), dtype=np.uint8)
        qa[i, :len(q)] = np.frombuffer(q.encode(), dtype=np.uint8) - 33
    return asc, qa, lens, np.asarray(bcs, dtype=np.int32).reshape(-1)


# ------------------------------------------------------------------------------------------------ compute
def count_graph_host(asc, quals, lens, bc, K=48, min_qual=7, min_freq=3, min_bc=2, device=0):
    """One call of the host-pointer C ABI (snk_count_graph); returns (off u64, bases u8) of the BVComp-ordered unitigs."""
    import ctypes as C
    from . import lib as _lib
    lib = _lib.load()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = lib.snk_ctx_create(device, C.byref(h), err, 512)
    if rc:
        raise _lib.SnkError(rc, err.value.decode(errors="replace"))
    try:
        asc = np.ascontiguousarray(asc, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
        bc = np.ascontiguousarray(bc, dtype=np.int32)
        r = _lib.SnkReads()
        r.n_reads, r.read_len = asc.shape[0], asc.shape[1]
        r.ascii, r.quals, r.lens, r.bc = asc.ctypes.data, quals.ctypes.data, lens.ctypes.data, bc.ctypes.data
        p = _lib.SnkParams()
        p.K, p.min_qual, p.min_freq, p.min_bc = K, min_qual, min_freq, min_bc
        out = _lib.SnkResult()
        rc = lib.snk_count_graph(h, C.byref(r), C.byref(p), C.byref(out), err, 512)
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))
        U = out.n_unitigs
        off = np.ctypeslib.as_array(out.unitig_off, shape=(U + 1,)).copy()
        tot = int(off[-1])
        bases = np.ctypeslib.as_array(out.unitig_bases, shape=(max(tot, 1),))[:tot].copy()
        stats = dict(n_instances=int(out.n_instances), n_kmers=int(out.n_kmers), n_unitigs=int(U))
        lib.snk_free(C.byref(out))
        return off, bases, stats
    finally:
        lib.snk_ctx_destroy(h)


class AsmSnGpu:
    """MartianStage (lib/tada/external/martian/src/lib.rs:407-411): split / main / join."""
    name = "asm_sn_gpu"

    def split(self, args):
        # one chunk: the whole job fits one MI355X (288 GB); mirrors MAIN_ASM_SN's single chunk (cmd_main_asm.rs:184-193)
        return {"chunks": [{"__mem_gb": 64, "__threads": 4}]}

    def main(self, args, outs, files_path="."):
        from . import graphio
        indexer = DeviceBcIndexer.from_file(args["barcode_whitelist"])      # no GPU: fails here, like every entry point
        try:
            asc, quals, lens, bc = read_fasth(args["fastqs"], indexer)
        finally:
            indexer.close()
        off, bases, stats = count_graph_host(asc, quals, lens, bc, K=48, min_qual=int(args.get("trim_min_qual", 7)),
                                             min_freq=int(args.get("min_kmer_obs", 3)), min_bc=2)
        path = outs.get("asm_graph") or str(Path(files_path) / "asm_graph.bv")
        graphio.write_bv(path, off, bases)
        outs = dict(outs)
        outs["asm_graph"] = path
        self.stats = stats
        return outs

    def join(self, args, outs, chunk_defs, chunk_outs):
        outs = dict(outs)
        outs["asm_graph"] = chunk_outs[0]["asm_graph"]
        return outs


STAGES = {AsmSnGpu.name: AsmSnGpu}


# ------------------------------------------------------------------------------------------------ protocol
class Metadata:
    def __init__(self, stage_name, stage_type, metadata_path, files_path, run_file):
        self.stage_name, self.stage_type = stage_name, stage_type
        self.metadata_path, self.files_path, self.run_file = metadata_path, files_path, run_file
        self._journalled = set()

    def path(self, name):
        return os.path.join(self.metadata_path, METADATA_PREFIX + name)

    def journal(self, name, force=False):
        jn = name if self.stage_type == "main" else f"{self.stage_type}_{name}"
        if jn in self._journalled and not force:
            return
        rf = f"{self.run_file}.{jn}"
        with open(rf + ".tmp", "w") as f:
            f.write(datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"))
        os.replace(rf + ".tmp", rf)
        self._journalled.add(jn)

    def write_raw(self, name, text):
        with open(self.path(name), "w") as f:
            f.write(text)
        self.journal(name)

    def write_json(self, name, obj):
        self.write_raw(name, json.dumps(obj, indent=2))

    def read_json(self, name):
        with open(self.path(name)) as f:
            return json.load(f)

    def log(self, level, message):
        with open(self.path("log"), "a") as f:
            f.write(f"{datetime.datetime.now().strftime('%Y-%m-%d %H:%M:%S')} [{level}] {message}\n")
        self.journal("log")

    def complete(self):
        self.write_raw("complete", datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"))


def martian_main(argv) -> int:
    """argv = [stage_name, split|main|join, metadata_path, files_path, run_file]"""
    md = Metadata(*argv[:5])
    md.log("time", "__start__")
    try:
        stage = STAGES[md.stage_name]()
        if md.stage_type == "split":
            md.write_json("stage_defs", stage.split(md.read_json("args")))
        elif md.stage_type == "main":
            outs = stage.main(md.read_json("args"), md.read_json("outs"), md.files_path)
            md.write_json("outs", outs)
        elif md.stage_type == "join":
            outs = stage.join(md.read_json("args"), md.read_json("outs"), md.read_json("chunk_defs"), md.read_json("chunk_outs"))
            md.write_json("outs", outs)
        else:
            raise ValueError(f"Unrecognized stage type {md.stage_type}")
        md.complete()
        return 0
    except BaseException:  # noqa: BLE001 -- the protocol wants every failure in _errors
        md.write_raw("errors", traceback.format_exc())
        return 1


if __name__ == "__main__":
    a = sys.argv[1:]
    if len(a) >= 6 and a[0] == "martian":
        sys.exit(martian_main(a[1:]))
    print("usage: python -m supernova_amd.martian martian <stage> <split|main|join> <metadata_path> <files_path> <run_file>", file=sys.stderr)
    sys.exit(2)
