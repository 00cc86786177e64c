"""Where does the sharded step spend host time?  (world 1, real RCCL group)"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
from supernova_amd.sharded import ShardedEngine
import supernova_amd.sharded as S
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
e = Engine(0)
sp = synth.synth_params(n, seed=0x5EED0001)
rows, quals, bc = e.synth(sp)
sh = ShardedEngine(e, dist)
import cProfile, pstats
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if i == 2:
        pr = cProfile.Profile(); pr.enable()
    res = sh.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
    if i == 2:
        pr.disable()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"step {i}: wall {1e3*(t1-t0):.1f} ms, events total {res.phase_ms['total']:.1f}", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
dist.destroy_process_group()
