"""MSP partition kernel time under the profiling switches (SNK_MSP_DBG): which of scan / slot atomics / record stores binds."""
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
for dbg in [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else '0,1,2'.split(','))]:
    e.set_option("msp_dbg", int(dbg))
    for rep in range(2):
        try:
            res = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, graph=False, sorted_table=False))
            print("dbg", dbg, "rep", rep, "msp kernel ms", round(res.kernel_ms["partition"], 2), "count", round(res.kernel_ms["count"], 2), flush=True)
        except Exception as ex:
            print("dbg", dbg, "failed:", str(ex)[:100], flush=True)
