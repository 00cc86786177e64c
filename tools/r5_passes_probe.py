"""bucket-range passes at bench size: the step with the one-pass partition and with 2 / 4 forced passes (same result, smaller slot array)"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
rm = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # repeat_mode (15: the repeat-rich genome of config.robust -- hot buckets in every pass)
e = Engine(0); sp = synth.synth_params(n, seed=0x5EED0001, **({'repeat_mode': rm} if rm else {})); rows, quals, bc = e.synth(sp)
ref = None
for passes in ("0", "2", "4", "0"):
    if passes == "0": e.clear_option("partition_passes")
    else: os.environ["SNK_PARTITION_PASSES"] = passes
    for _ in range(3):
        r = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
    sig = (r.n_kmers, r.n_unitigs, r.unitig_total_bases, r.n_instances, r.n_supermers)
    ref = ref or sig
    print("passes", e.last_partition_passes(), {k: round(v, 1) for k, v in r.phase_ms.items()}, "arena GB", round(r.scratch_bytes / 2**30, 1), "hot buckets", r.n_hot_buckets, "same result:", sig == ref, flush=True)
