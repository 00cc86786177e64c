import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import goldens, oracle_lib
from supernova_amd.engine import Engine, Params
c = goldens.load("adversarial")
dev = torch.device("cuda", 0)
rows = torch.from_numpy(c.rows.view(np.int32)).to(dev); quals = torch.from_numpy(np.ascontiguousarray(c.quals)).to(dev)
bc = torch.from_numpy(c.bc.astype(np.int32)).to(dev); lens = torch.from_numpy(c.lens.astype(np.uint16).view(np.int16)).to(dev)
gl = c.exp_goodlens
for fresh in (True, False):
    e = Engine(0)
    for min_bc, bcarg in [(2, None), (0, bc), (1, bc), (2, bc), (2, None)]:
        if fresh:
            e.close(); e = Engine(0)
        res = e.count_graph(rows, c.read_len, quals=quals, bc=bcarg, lens=lens, params=Params(K=48, min_bc=min_bc), ign_bc_below=c.ign_bc_below)
        o = oracle_lib.OracleResult(c.codes, gl, None if bcarg is None else c.bc, min_bc=min_bc, ign_bc_below=c.ign_bc_below, hbv=False)
        k = res.keys()
        same = k.shape[0] == o.keys.shape[0] and np.array_equal(k, o.keys)
        print("fresh" if fresh else "reuse", "min_bc", min_bc, "bc", bcarg is not None, "gpu", res.n_kmers, "oracle", len(o.keys), "OK" if same else "MISMATCH", "split", res.buckets_split, "NB", res.n_buckets)
