"""W simulated ranks on ONE GPU at bench-like sizes: per-rank phase timings of the sharded path (the rank-0
join is the serial part).  usage: python tools/sim_scale.py W reads_per_rank [reps] [serial]
(serial: the ranks compute one after the other between exchanges, so every phase time is that of a rank alone on the GPU)"""
import sys, threading, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
from supernova_amd.sharded import ShardedEngine, SimWorld

W = int(sys.argv[1]); per = int(float(sys.argv[2])); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
world = SimWorld(W, serial=len(sys.argv) > 4 and sys.argv[4] == 'serial')
sp = synth.synth_params(W * per, seed=0x5EED0002)
out = [None] * W
errs = []

def worker(r):
    try:
        torch.cuda.set_device(0)
        e = Engine(0)
        rows, quals, bc = e.synth(sp, first=r * per, n=per)
        c = world.comm(r)
        sh = ShardedEngine(e, c)
        for rep in range(reps):
            torch.cuda.synchronize(); world.barrier_obj.wait(); t0 = time.time()
            if r == 0: world.turn_of = 0
            c._begin_section()
            res = sh.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=Params(K=48), read_index_base=r * per)
            c._end_section()
            torch.cuda.synchronize(); t1 = time.time()
            out[r] = (t1 - t0, res.phase_ms, res.n_kmers, res.n_frags, res.n_queries, res.n_unitigs, res.n_instances)
            jm = getattr(res, "join_ms", {})
            if r == 0 and jm:
                comp = sum(v for k, v in jm.items() if not k.endswith("(x)"))
                print(f"rep{rep} rank0 join sections (ms; (x) = exchange incl. waiting for the other ranks): "
                      + " ".join(f"{k}={v:.1f}" for k, v in jm.items()) + f" | compute {comp:.1f}", flush=True)
            world.barrier_obj.wait()
            if r == 0:
                for q in range(W):
                    w, ph, nk, nf, nq, nu, ni = out[q]
                    print(f"rep{rep} rank{q} wall={w*1e3:.0f}ms inst={ni} kmers={nk} frags={nf} queries={nq} unitigs={nu} "
                          + " ".join(f"{k}={v:.0f}" for k, v in ph.items()), flush=True)
            world.barrier_obj.wait()
    except BaseException as ex:
        errs.append(ex); world.barrier_obj.abort(); raise

ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
[t.start() for t in ts]; [t.join() for t in ts]
if errs: raise errs[0]
