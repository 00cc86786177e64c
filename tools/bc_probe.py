"""f3: device barcode ids (snk_dev_bc_ids) -- time of one lookup pass over n barcode fields against a whitelist of w lines.
usage: python tools/bc_probe.py [n_fields] [whitelist_lines]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ctypes as C
import numpy as np
import torch
from supernova_amd.martian import DeviceBcIndexer
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
w = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_700_000
rng = np.random.default_rng(1)
codes = rng.integers(0, 4, (w, 16), dtype=np.uint8)
wl = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
text = np.concatenate([wl, np.full((w, 1), 10, dtype=np.uint8)], axis=1).tobytes()
t0 = time.perf_counter(); ix = DeviceBcIndexer(text); t1 = time.perf_counter()
print(f"whitelist {w} lines: index built in {t1 - t0:.2f} s (host sort + upload)")
dev = torch.device("cuda", 0)
pick = torch.randint(0, w, (n,), device=dev)
f = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
f[:, :16] = torch.from_numpy(wl).to(dev)[pick]
f[:, 16] = ord("-"); f[:, 17] = ord("1")
f[::7, 3] = ord("N")                      # every 7th field is not on the whitelist
ids = torch.empty((n,), dtype=torch.int32, device=dev)
err = C.create_string_buffer(256)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = ix.lib.snk_dev_bc_ids(ix._ctx, ix._ix, f.data_ptr(), 32, n, ids.data_ptr(), None, err, 256)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert rc == 0, err.value
    print(f"rep {rep}: {n} fields in {dt * 1e3:.2f} ms = {n / dt / 1e9:.2f} G fields/s, hits {int((ids > 0).sum())}")
