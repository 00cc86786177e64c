"""Read paths and duplicate marks against the reference binary itself (oracle/_ref/snref_driver) on the GPU box, at sizes and error
rates beyond the committed test (tests/test_gpu_parity.py::test_vs_reference_binary_200k): the whole of pathReads + MarkDups, bit for
bit.  Test infrastructure (it runs the reference).  usage: python tests/tools/path_parity_vs_ref.py [n_reads=1000000] [sub_ppm=6000] [seed]"""
import math, os, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import refio
from supernova_amd import synth
from supernova_amd.engine import Engine, Params

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ppm = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x5EED0A11
sp = synth.synth_params(n, seed=seed, sub_ppm=ppm)
lam, term, cum = 150 * ppm / 1e6, math.exp(-150 * ppm / 1e6), 0.0
for j in range(4):
    cum += term; sp.err_cdf[j] = min(0xFFFFFFFF, int(cum * 4294967296.0)); term *= lam / (j + 1)
rows, quals, bc = synth.synth_host(sp)
asc = synth.codes_to_ascii(synth.unpack_rows(rows, 150))
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    refio.write_snkrd(Path(td) / "in.snkrd", np.full(n, 150), asc, quals, bc)
    t0 = time.time()
    refio.run_ref(Path(td) / "in.snkrd", Path(td) / "out", threads=min(64, os.cpu_count() or 8))
    print(f"reference: {time.time() - t0:.1f} s", flush=True)
    d = refio.read_ref_dump(Path(td) / "out")
dev = torch.device("cuda", 0)
e = Engine(0)
rows_d, quals_d, bc_d = torch.from_numpy(rows.view(np.int32)).to(dev), torch.from_numpy(quals).to(dev), torch.from_numpy(bc).to(dev)
res = e.count_graph(rows_d, 150, quals=quals_d, bc=bc_d, params=Params(K=48))
us = res.unitigs()
ok_u = us == d["unitigs"]
off, ne, edges, info = res.path_reads(rows_d, 150, quals_d, mark_dups=True, bc=bc_d)
ok_p = np.array_equal(ne.astype(np.int32), d["path_n"]) and np.array_equal(edges, d["path_edges"]) and np.array_equal(off, d["path_off"])
ok_d = np.array_equal(info["dups"]["dup"], d["dup"]) and info["dups"]["interdup_rate"] == d["interdup"]
print(f"{n} reads, {ppm / 1e4:.2f} % errors: {len(us)} unitigs {'==' if ok_u else '!='} reference; paths ({int((ne > 0).sum())} placed, {int((ne > 1).sum())} with several edges, "
      f"second pass {info['n_slow']} reads) {'==' if ok_p else '!='} reference; dup marks ({int(d['dup'].sum())} pairs) {'==' if ok_d else '!='} reference")
sys.exit(0 if (ok_u and ok_p and ok_d) else 1)
