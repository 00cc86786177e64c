#!/usr/bin/env python3
"""SHA-256 fixtures of whole runs of the REFERENCE ITSELF at sizes whose dumps are too large to commit
(BASELINE config 1 = 10 M x 150 bp, seed 0x5EED0001).

Build container only (needs oracle/_ref/snref_driver[60], i.e. /root/reference compiled by oracle/ref/build_ref.sh).
The inputs are the seeded synthetic reads of libsnk's generator (data); the digests (tests/bighash.py) of the reference's
dump -- good lengths, retained table (keys, counts, pruned contexts), spectrum, canonical unitigs -- go to
tests/golden/big_hashes.json together with the reference's own summary line and its wall time.

usage: python tests/golden/make_big_hashes.py [case ...]      cases: c1_10m  c1_10m_k60  c1_2m  c1_2m_k60  robust_*
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import bighash  # noqa: E402
import refio  # noqa: E402
from supernova_amd import synth  # noqa: E402

OUT = Path(__file__).resolve().parent / "big_hashes.json"

# name -> (n_reads, seed, K[, generator overrides])
CASES = {
    "c1_10m": (10_000_000, 0x5EED0001, 48),
    "c1_10m_k60": (10_000_000, 0x5EED0001, 60),
    "c1_2m": (2_000_000, 0x5EED0001, 48),
    "c1_2m_k60": (2_000_000, 0x5EED0001, 60),
    # off the bench's operating point (bench.py config.robust runs the same models at 100 M reads): more sequencing errors, half the
    # coverage, a repeat-rich genome (interspersed families, segmental duplications, STRs, poly-A: csrc/snk_synth.h)
    "robust_err06_200k": (200_000, 0x5EED0206, 48, dict(sub_ppm=6000)),
    "robust_err15_200k": (200_000, 0x5EED0215, 48, dict(sub_ppm=15000, lowq_tail_ppm=500000)),
    "robust_cov28_200k": (200_000, 0x5EED0228, 48, dict(genome_len=200_000 * 150 // 28)),
    "robust_repeats_200k": (200_000, 0x5EED0201, 48, dict(repeat_mode=15)),
    "robust_repeats_1m": (1_000_000, 0x5EED0202, 48, dict(repeat_mode=15)),
    "robust_repeats_200k_k60": (200_000, 0x5EED0201, 60, dict(repeat_mode=15)),
}


def make(name: str) -> dict:
    n, seed, K = CASES[name][:3]
    ov = CASES[name][3] if len(CASES[name]) > 3 else {}
    sp = synth.synth_params(n, seed=seed, **ov)
    rows, quals, bc = synth.synth_host(sp)
    asc = synth.codes_to_ascii(synth.unpack_rows(rows, sp.read_len))
    del rows
    threads = os.cpu_count()
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        td = Path(td)
        refio.write_snkrd(td / "in.snkrd", np.full(n, sp.read_len), asc, quals, bc)
        del asc, quals
        t0 = time.time()
        log = refio.run_ref(td / "in.snkrd", td / "out", threads=threads, K=K, timeout=6 * 3600)
        secs = time.time() - t0
        d = refio.read_ref_dump(td / "out", K=K)
    summary = [l for l in log.splitlines() if l.startswith("SNREF_DUMP")][-1]
    if d["hist"] is not None:
        hist = np.asarray(d["hist"]["vals"], dtype=np.int64)
    else:       # the K=60 variant writes no spectrum file: the histogram of the retained counts stands in
        hist = np.bincount(d["kmers"]["count"]).astype(np.int64)
    dg = bighash.digest(d["goodlens"], d["kmers"]["k"], d["kmers"]["count"], d["kmers"]["ctx"], d["unitigs"], hist,
                        kw=3 if K == 48 else 4)
    if ov:
        dg["overrides"] = ov
    dg.update(K=K, seed=seed, ref_summary=summary, ref_dump_seconds=round(secs, 1), ref_threads=threads,
              hist_from="reference json" if d["hist"] is not None else "retained counts")
    return dg


if __name__ == "__main__":
    names = sys.argv[1:] or ["c1_2m", "c1_2m_k60", "c1_10m", "c1_10m_k60"]
    allh = json.loads(OUT.read_text()) if OUT.exists() else {}
    for nm in names:
        allh[nm] = make(nm)
        OUT.write_text(json.dumps(allh, indent=1, sort_keys=True) + "\n")
        print(nm, allh[nm]["ref_summary"], f'{allh[nm]["ref_dump_seconds"]} s', flush=True)
