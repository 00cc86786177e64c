"""Host-pointer ABI (snk_count_graph) at a few million reads: PCIe-inclusive wall time, and the unitigs against the device path."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from supernova_amd import synth, martian
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
sp = synth.synth_params(n, seed=0x5EED0333)
rows, quals, bc = synth.synth_host(sp)
asc = synth.codes_to_ascii(synth.unpack_rows(rows, 150))
t0 = time.time()
off, bases, stats = martian.count_graph_host(asc, quals[:, :150].copy(), np.full(n, 150, np.uint16), bc)
t1 = time.time()
print(f"host API: {n} reads, {stats}, wall {t1 - t0:.2f} s -> {stats['n_instances'] / (t1 - t0) / 1e9:.2f} Gk-mers/s PCIe + host ordering inclusive", flush=True)
e = Engine(0)
dev = torch.device("cuda", 0)
res = e.count_graph(torch.from_numpy(rows.view(np.int32)).to(dev), 150, quals=torch.from_numpy(quals).to(dev), bc=torch.from_numpy(bc).to(dev))
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
us = sorted((lut[bases[int(off[i]):int(off[i + 1])]].tobytes().decode() for i in range(len(off) - 1)), key=lambda s: (-len(s), s))
print("unitigs equal device path:", us == res.unitigs())
