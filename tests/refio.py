"""Test-side I/O for the executable reference oracle (oracle/_ref/snref_driver) and golden fixtures.

SNKRD001 input file (little endian): magic[8], u64 n, u32 stride, u32 has_bc, i64 ign_bc_below,
u16 len[n], u8 bases_ascii[n*stride], u8 quals[n*stride], i32 bc[n].
"""
from __future__ import annotations

import json
import struct
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
REF_DRIVER = ROOT / "oracle" / "_ref" / "snref_driver"
GOLDEN = ROOT / "tests" / "golden"

KREC = np.dtype([("k", "<u4", 3), ("count", "<u4"), ("ctx", "u1"), ("pad", "u1", 3)])
KREC60 = np.dtype([("k", "<u4", 4), ("count", "<u4"), ("ctx", "u1"), ("pad", "u1", 3)])
REF_DRIVER60 = ROOT / "oracle" / "_ref" / "snref_driver60"


def write_snkrd(path, lens, bases_ascii, quals, bc=None, ign_bc_below=0):
    n, stride = bases_ascii.shape
    with open(path, "wb") as f:
        f.write(b"SNKRD001")
        f.write(struct.pack("<QIIq", n, stride, 1 if bc is not None else 0, ign_bc_below))
        f.write(np.asarray(lens, dtype="<u2").tobytes())
        f.write(np.ascontiguousarray(bases_ascii, dtype=np.uint8).tobytes())
        f.write(np.ascontiguousarray(quals, dtype=np.uint8).tobytes())
        if bc is not None:
            f.write(np.asarray(bc, dtype="<i4").tobytes())


def run_ref(snkrd, outdir, threads=8, mode="dump", min_qual=7, min_freq=3, min_bc=2, timeout=3600, K=48):
    outdir = Path(outdir)
    outdir.mkdir(parents=True, exist_ok=True)
    r = subprocess.run([str(REF_DRIVER if K == 48 else REF_DRIVER60), str(snkrd), str(outdir), str(threads), mode, str(min_qual), str(min_freq),
                        str(min_bc)], capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"snref_driver failed ({r.returncode}):\n{r.stdout[-4000:]}\n{r.stderr[-4000:]}")
    return r.stdout


def read_ref_dump(outdir, K=48):
    outdir = Path(outdir)
    out = {}
    out["goodlens"] = np.fromfile(outdir / "goodlens.u32", dtype="<u4")
    out["kmers"] = np.fromfile(outdir / "kmers.bin", dtype=KREC if K == 48 else KREC60)
    out["unitigs"] = (outdir / "unitigs.txt").read_text().split()
    out["hbv"] = (outdir / "hbv.txt").read_text()
    hist = outdir / "stats" / "histogram_kmer_count.json"
    out["hist"] = json.loads(hist.read_text()) if hist.exists() else None
    # f1/f2: read paths (K=48 dumps), the graph file and its involution as DF writes them
    pp = outdir / "paths.txt"
    if pp.exists():
        offs, ns, edges = [], [], []
        for line in pp.read_text().splitlines():
            t = line.split()
            offs.append(int(t[0])); ns.append(int(t[1])); edges.extend(int(x) for x in t[2:])
        out["path_off"] = np.asarray(offs, dtype=np.int32)
        out["path_n"] = np.asarray(ns, dtype=np.int32)
        out["path_edges"] = np.asarray(edges, dtype=np.int32)
    # f4: MarkDups over those paths -- inter-barcode duplicate rate, logged artifactual-duplicate percentage, flag per pair
    md = outdir / "markdups.txt"
    out["dup"] = None
    if md.exists():
        head, bits = md.read_text().split("\n")[:2]
        t = head.split()
        out["interdup"] = float(t[0])
        out["art_perc"] = float(t[1]) if t[1] not in ("nan", "-nan") else 0.0
        out["dup"] = np.frombuffer(bits.encode(), dtype=np.uint8) - ord("0")
        assert len(out["dup"]) == int(t[2])
    for nm in ("a.hbv", "a.inv"):
        f = outdir / nm
        out[nm] = np.frombuffer(f.read_bytes(), dtype=np.uint8) if f.exists() else None
    return out
