"""SHA-256 fingerprints of whole-run results, for parity at sizes whose dumps are too large to commit (SURVEY.md 8(c):
"hash large ones").  The same functions digest the reference's dump (tests/golden/make_big_hashes.py, build container)
and the HIP path's result (tests/test_gpu_bigparity.py, GPU box), so equal digests = bit-equal results.

Canonical forms:
  goodlens  u32 LE per read, read order
  keys      retained canonical k-mers in ascending key order, KW u32 LE words each (3 at K=48, 4 at K=60)
  counts    u32 LE per retained k-mer, same order, saturated at 2^24-1 (kmers/ReadPather.h:127-131,145)
  ctx       u8 per retained k-mer, same order (pred<<4 | succ, KMerContext.h:27-28)
  unitigs   canonical unitig strings (ACGT), sorted as byte strings, joined by '\n' (+ trailing '\n')
  hist      k-mer spectrum, i64 LE, bins 0..max observed count (trailing zero bins dropped)
"""
from __future__ import annotations

import hashlib

import numpy as np


def _h(b) -> str:
    return hashlib.sha256(b).hexdigest()


def digest(goodlens, keys, counts, ctx, unitigs, hist, kw=3) -> dict:
    keys = np.ascontiguousarray(np.asarray(keys)[:, :kw], dtype="<u4")
    hist = np.asarray(hist, dtype=np.int64)
    nz = np.nonzero(hist)[0]
    hist = hist[: (nz[-1] + 1 if len(nz) else 0)]
    us = sorted(u.encode() if isinstance(u, str) else bytes(u) for u in unitigs)
    hu = hashlib.sha256()
    for u in us:
        hu.update(u)
        hu.update(b"\n")
    return {
        "n_reads": int(len(goodlens)),
        "n_kmers": int(keys.shape[0]),
        "n_unitigs": int(len(us)),
        "unitig_bases": int(sum(len(u) for u in us)),
        "goodlens": _h(np.ascontiguousarray(goodlens, dtype="<u4").tobytes()),
        "keys": _h(keys.tobytes()),
        "counts": _h(np.minimum(np.asarray(counts, dtype=np.uint64), (1 << 24) - 1).astype("<u4").tobytes()),
        "ctx": _h(np.ascontiguousarray(ctx, dtype=np.uint8).tobytes()),
        "unitigs": hu.hexdigest(),
        "hist": _h(hist.astype("<i8").tobytes()),
    }
