"""Per-group (per-barcode) local graphs -- BASELINE config 5 / SURVEY.md 8(e) "C5".

One GPU: `Engine.count_graph(..., group=..., params=Params(grouped=True, min_bc=0))` counts, prunes and walks every
(group, k-mer) in one batched run (the group id rides in the 32 low bits of the 128-bit key, free at K=48).
Several GPUs: the groups are independent units, so the path does not need a collective -- replicas only.  Groups are
dealt to ranks as contiguous ranges balanced by read count (the reference selects read ranges per barcode through the
`bci` partition, lib/assembly/src/10X/DF.cc:464-469, 10X/MakeLocalsTools.cc:102)."""
from __future__ import annotations

import numpy as np


def partition_groups(reads_per_group: np.ndarray, world: int) -> np.ndarray:
    """Contiguous group ranges per rank, balanced by cumulative read count.
    Returns bounds[world+1]: rank r owns groups [bounds[r], bounds[r+1])."""
    n = np.asarray(reads_per_group, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(n)])
    total = int(cum[-1])
    bounds = np.zeros(world + 1, dtype=np.int64)
    for r in range(1, world):
        bounds[r] = int(np.searchsorted(cum, total * r / world, side="left"))
    bounds[world] = len(n)
    return np.maximum.accumulate(bounds)


def read_slab(bci_offsets: np.ndarray, bounds: np.ndarray, rank: int) -> tuple[int, int]:
    """Reads [first, last) of a rank given the `bci` table (first read of every group, plus the total at the end)."""
    return int(bci_offsets[bounds[rank]]), int(bci_offsets[bounds[rank + 1]])
