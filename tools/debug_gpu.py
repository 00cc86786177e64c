"""Diagnostics on a GPU box: where does the HIP path diverge from the golden vectors?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import goldens
from supernova_amd.engine import Engine, Params

name = sys.argv[1] if len(sys.argv) > 1 else "synth_2k_err"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = goldens.load(name)
dev = torch.device("cuda", 0)
e = Engine(0)
res = e.count_graph(torch.from_numpy(c.rows.view(np.int32)).to(dev), c.read_len,
                    quals=torch.from_numpy(np.ascontiguousarray(c.quals)).to(dev),
                    bc=torch.from_numpy(c.bc.astype(np.int32)).to(dev),
                    lens=torch.from_numpy(c.lens.astype(np.uint16).view(np.int16)).to(dev),
                    params=Params(K=48, n_buckets=nb), ign_bc_below=c.ign_bc_below)
print("n_inst", res.n_instances, "n_super", res.n_supermers, "NB", res.n_buckets, "n_kmers", res.n_kmers, "exp", len(c.exp_keys),
      "split", res.buckets_split, "maxslots", res.max_slots_used, "unitigs", res.n_unitigs, "circles", res.n_circles, "rounds", res.rank_rounds)
print("phases", res.phase_ms)
k = res.keys()
print("word3 zero:", np.all(k[:, 3] == 0))
tk = [tuple(x) for x in k[:, :3]]
ek = [tuple(x) for x in c.exp_keys]
print("sorted:", tk == sorted(tk), "distinct:", len(set(tk)))
sg, se = set(tk), set(ek)
print("common", len(sg & se), "only_gpu", len(sg - se), "only_exp", len(se - sg))
cnt = dict(zip(tk, res.counts())); ecnt = dict(zip(ek, c.exp_counts))
bad = [(x, cnt[x], ecnt[x]) for x in sg & se if cnt[x] != ecnt[x]]
print("count mismatches among common:", len(bad), bad[:5])
cx = dict(zip(tk, res.ctx())); ecx = dict(zip(ek, c.exp_ctx))
badc = [(x, cx[x], ecx[x]) for x in sg & se if cx[x] != ecx[x]]
print("ctx mismatches among common:", len(badc), badc[:5])
if tk == ek:
    u = res.unitigs()
    print("unitigs equal:", u == c.exp_unitigs, len(u), len(c.exp_unitigs))
    if u != c.exp_unitigs:
        su, sx = set(u), set(c.exp_unitigs)
        print("common unitigs", len(su & sx), "only gpu", len(su - sx), "only exp", len(sx - su))
        for x in list(su - sx)[:3]: print("GPU ", len(x), x[:80])
        for x in list(sx - su)[:3]: print("EXP ", len(x), x[:80])
