"""Count kernel time under its profiling switches (SNK_COUNT_DBG; results invalid for dbg != 0):
   1 = roll + canonicalise + hash only (no table), 2 = probe/claim but no count/context/barcode updates, 4 = no supermer de-duplication."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
e = Engine(0)
sp = synth.synth_params(n, seed=0x5EED0001)
rows, quals, bc = e.synth(sp)
modes = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (0, 1, 2, 4)
for dbg in modes:
    e.set_option("count_dbg", int(dbg))
    for rep in range(2):
        try:
            res = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, graph=False, sorted_table=False))
            msg = f"count kernel {res.kernel_ms['count']:.1f} ms, partition kernel {res.kernel_ms['partition']:.1f} ms, retained {res.n_kmers}, split {res.buckets_split}, max slots {res.max_slots_used}"
        except Exception as ex:
            msg = "failed: " + str(ex)[:80]
    print("dbg", dbg, msg, flush=True)
