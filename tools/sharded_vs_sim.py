"""The sharded step over a real one-rank RCCL group vs the simulated transport; every exchange is compared byte for byte."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
from supernova_amd.sharded import ShardedEngine, SimWorld, TorchComm
n = int(float(sys.argv[1]))
e = Engine(0)
sp = synth.synth_params(n, seed=0x5EED0001)
rows, quals, bc = e.synth(sp)
class Spy(TorchComm):
    def all_to_all_v(self, send, send_counts, alloc=None):
        recv, rc = super().all_to_all_v(send, send_counts, alloc)
        torch.cuda.synchronize()
        same = recv.numel() == send.numel() and bool(torch.equal(recv, send))
        print("  a2a", send.numel(), recv.numel(), rc, "equal" if same else "DIFFERENT", flush=True)
        return recv, rc
for name, comm in (("sim", SimWorld(1).comm(0)), ("torch", Spy(dist))):
    sh = ShardedEngine(e, comm)
    res = sh.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48))
    print(name, res.n_kmers, res.n_unitigs, res.n_supermers, res.n_frags, flush=True)
dist.destroy_process_group()
