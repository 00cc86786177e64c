"""Large parity run on the GPU box: HIP path vs the reference binary itself (oracle/_ref/snref_driver) on a seeded
synthetic workload of the bench's model.  Bit-exact table (key, count, context) and unitigs.
usage: python tests/tools/parity_vs_ref.py [n_reads=1000000] [error_free]"""
import sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import refio
from supernova_amd import synth
from supernova_amd.engine import Engine, Params

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ef = len(sys.argv) > 2
sp = synth.synth_params(n, seed=0x5EED0042, error_free=ef)
rows, quals, bc = synth.synth_host(sp)
asc = synth.codes_to_ascii(synth.unpack_rows(rows, 150))
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    refio.write_snkrd(Path(td) / "in.snkrd", np.full(n, 150), asc, quals, bc)
    t0 = time.time()
    import os
    refio.run_ref(Path(td) / "in.snkrd", Path(td) / "out", threads=os.cpu_count())
    print(f"reference dump: {time.time()-t0:.1f} s")
    d = refio.read_ref_dump(Path(td) / "out")
dev = torch.device("cuda", 0)
e = Engine(0)
res = e.count_graph(torch.from_numpy(rows.view(np.int32)).to(dev), 150, quals=torch.from_numpy(quals).to(dev),
                    bc=torch.from_numpy(bc).to(dev), params=Params(K=48))
k = res.keys()
ok = [np.array_equal(res.good_len().astype(np.uint32), d["goodlens"]),
      k.shape[0] == len(d["kmers"]) and np.array_equal(k[:, :3], d["kmers"]["k"]),
      np.array_equal(np.minimum(res.counts(), (1 << 24) - 1), d["kmers"]["count"]),
      np.array_equal(res.ctx(), d["kmers"]["ctx"]),
      res.unitigs() == d["unitigs"]]
print(f"n_reads={n} instances={res.n_instances} retained={res.n_kmers} unitigs={res.n_unitigs} "
      f"goodlen/keys/counts/ctx/unitigs equal: {ok}")
sys.exit(0 if all(ok) else 1)
