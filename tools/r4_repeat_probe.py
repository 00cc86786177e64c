"""Which structure of the repeat-rich genome costs what (round 4): the step on 1e7 / 1e8 reads with one structure at a time.
usage: python tools/r4_repeat_probe.py [n_reads]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
eng = Engine(0)
for name, mode in (("none", 0), ("families", 1), ("segdups", 2), ("STRs", 4), ("polyA", 8), ("all", 15)):
    sp = synth.synth_params(n, seed=0x5EED0042, repeat_mode=mode)
    rows, quals, bc = eng.synth(sp)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        r = eng.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
    print(f"{name}: {dt:.1f} ms | " + " ".join(f"{k} {v:.1f}" for k, v in r.phase_ms.items() if k in ("partition", "count", "graph")) +
          f" | buckets {r.n_buckets} split {r.buckets_split} overflow {r.n_overflow} kmers {r.n_kmers} unitigs {r.n_unitigs} max_slots {r.max_slots_used}", flush=True)
    if name in ("STRs", "polyA", "all") and len(sys.argv) > 2:
        # where the partition's time goes (results invalid): 3 = the scan alone, 2 = no slot atomics, 1 = no record stores
        for dbg in ("3", "2", "1"):
            eng.set_option("msp_dbg", int(dbg))
            try:
                for rep in range(2):
                    r2 = eng.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
                print(f"   SNK_MSP_DBG={dbg}: partition {r2.phase_ms['partition']:.1f} ms", flush=True)
            except Exception as ex:
                print(f"   SNK_MSP_DBG={dbg}: {type(ex).__name__} {str(ex)[:100]}", flush=True)
            eng.clear_option("msp_dbg")
    del rows, quals, bc, r
