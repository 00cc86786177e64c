"""W in-process ranks on ONE GPU at bench-like sizes: per-rank phase timings of snk_shard_step (the ranks share the GPU, so the
phase times are upper bounds of what a rank alone would take; the exchanges are device copies).
usage: python tools/sim_scale.py W reads_per_rank [reps] [repeat_mode]      (repeat_mode 15: the repeat-rich genome of config.robust -- hot buckets on their owners)"""
import sys, threading, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
from supernova_amd.sharded import ShardedEngine, SimWorld

W = int(sys.argv[1]); per = int(float(sys.argv[2])); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
torch.cuda.init()
DEV_MB = torch.cuda.mem_get_info(0)[1] >> 20
world = SimWorld(W)
bar = threading.Barrier(W)
rmode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sp = synth.synth_params(W * per, seed=0x5EED0002, **({'repeat_mode': rmode} if rmode else {}))
out = [None] * W
pairs = [None] * W
errs = []

def worker(r):
    try:
        torch.cuda.set_device(0)
        e = Engine(0)
        # a simulated rank plans as the rank it stands for would: with a device of its own (the W contexts share this one, and a context's
        # plans divide what it finds free: snk_ctx_plan_mem)
        e.set_option("plan_mem_mb", DEV_MB)
        rows, quals, bc = e.synth(sp, first=r * per, n=per)
        sh = ShardedEngine(e, world.comm(r))
        for rep in range(reps):
            torch.cuda.synchronize(); bar.wait(); t0 = time.time()
            res = sh.count_graph(rows, sp.read_len, quals=quals, bc=bc, params=Params(K=48), read_index_base=r * per, total_reads=W * per)
            torch.cuda.synchronize(); t1 = time.time()
            out[r] = (t1 - t0, res.phase_ms, res.join_ms, res.n_kmers, res.n_frags, res.n_queries, res.n_unitigs, res.n_instances, res.host_syncs, res.exchange_bytes, int(res.raw.n_hot_buckets))
            pairs[r] = res.pair_max_bytes
            bar.wait()
            if r == 0 and W > 1:
                # link balance: per exchange, what the ranks put on the wire in all, the mean per (sender, receiver) pair and the fullest pair
                print(f"rep{rep} pair balance (MB): " + " | ".join(
                    f"{k}: total {sum(out[q][9][k] for q in range(W)) / 1e6:.0f} mean/pair {sum(out[q][9][k] for q in range(W)) / (W * (W - 1)) / 1e6:.1f} max pair {max(pairs[q][k] for q in range(W)) / 1e6:.1f}"
                    for k in pairs[0]), flush=True)
            if r == 0:
                for q in range(W):
                    w, ph, jm, nk, nf, nq, nu, ni, hs, xb, nh = out[q]
                    print(f"rep{rep} rank{q} wall={w*1e3:.0f}ms hot_buckets={nh} inst={ni} kmers={nk} frags={nf} queries={nq} unitigs={nu} read-backs={hs} "
                          + " ".join(f"{k}={v:.0f}" for k, v in ph.items()) + " | join: " + " ".join(f"{k}={v:.1f}" for k, v in jm.items())
                          + f" | sent MB: " + " ".join(f"{k}={v/1e6:.0f}" for k, v in xb.items()), flush=True)
            bar.wait()
    except BaseException as ex:
        errs.append(ex); world.barrier_obj.abort(); bar.abort(); raise

ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
[t.start() for t in ts]; [t.join() for t in ts]
if errs: raise errs[0]
