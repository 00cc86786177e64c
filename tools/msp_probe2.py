"""Why is the partition kernel faster in grouped runs?  Variants of one 100 M-read call."""
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
def run(tag, **kw):
    for rep in range(2):
        res = e.count_graph(rows, 150, quals=quals, **kw)
    print(f"{tag:40s} msp {res.kernel_ms['partition']:.1f} ms  count {res.kernel_ms['count']:.1f}  NB {res.n_buckets} super {res.n_supermers} ovf {res.n_overflow} kmers {res.n_kmers}", flush=True)
run("ungrouped, bc rule", bc=bc, params=Params(graph=False, sorted_table=False))
run("ungrouped, no bc array", bc=None, params=Params(graph=False, sorted_table=False))
run("grouped (bucket target 900)", bc=None, group=bc, params=Params(graph=False, sorted_table=False, grouped=True, min_bc=0))
e.set_option("target_inst", 4000)
run("grouped, bucket target 4000", bc=None, group=bc, params=Params(graph=False, sorted_table=False, grouped=True, min_bc=0))
zero = torch.zeros_like(bc)
run("grouped, one group, target 4000", bc=None, group=zero, params=Params(graph=False, sorted_table=False, grouped=True, min_bc=0))
