"""The drop-in host seam end to end: host arrays (packed rows, quality rows, barcodes) -> snk_count_graph -> .bv file image, timed
(PCIe, the chunked trim, the device step, the BVComp order and the .bv packing on the device, the download, the file write all
inside), next to the device-resident step.  usage: python tools/host_seam_probe.py [n_reads=1e8] [pinned|pageable] [reps=3]"""
import ctypes as C, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from supernova_amd import synth, lib as _lib, graphio
from supernova_amd.engine import Engine, Params

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "pinned"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lib = _lib.load()
e = Engine(0)
sp = synth.synth_params(n, seed=0x5EED0001)
rows_d, quals_d, bc_d = e.synth(sp, qstride=150)
res = e.count_graph(rows_d, 150, quals=quals_d, bc=bc_d, params=Params(K=48))
dev_ms = res.phase_ms["total"]
ref_units = res.unitigs() if n <= 20_000_000 else None
n_inst = res.n_instances
err = C.create_string_buffer(512)


def host_array(t):
    """device tensor -> host numpy array in pinned or pageable memory"""
    nb = t.numel() * t.element_size()
    if mode == "pinned":
        p = C.c_void_p()
        _lib.check(lib.snk_host_alloc_pinned(nb, C.byref(p), err, 512))
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nb,))
    else:
        a = np.empty(nb, dtype=np.uint8)
    e._download(t.data_ptr(), a.ctypes.data, nb)
    return a


rows_h, quals_h, bc_h = host_array(rows_d), host_array(quals_d), host_array(bc_d)
del rows_d, quals_d, bc_d, res
e.close()
torch.cuda.empty_cache()
h = C.c_void_p()
_lib.check(lib.snk_ctx_create(0, C.byref(h), err, 512))
r = _lib.SnkReads()
r.n_reads, r.read_len = n, 150
r.rows, r.quals, r.bc = rows_h.ctypes.data, quals_h.ctypes.data, bc_h.ctypes.data
p = _lib.SnkParams()
p.K, p.min_qual, p.min_freq, p.min_bc, p.flags = 48, 7, 3, 2, 16 | 32          # SNK_F_NO_TABLE | SNK_F_BV_IMAGE
out_path = "/tmp/seam_probe.bv"
for rep in range(reps):
    out = _lib.SnkResult()
    t0 = time.perf_counter()
    rc = lib.snk_count_graph(h, C.byref(r), C.byref(p), C.byref(out), err, 512)
    assert rc == 0, err.value
    img = np.ctypeslib.as_array(out.bv_image, shape=(int(out.bv_bytes),))
    with open(out_path, "wb") as f:
        f.write(img.data)
    t1 = time.perf_counter()
    gb = (rows_h.nbytes + quals_h.nbytes + bc_h.nbytes) / 1e9
    print(f"rep {rep}: {mode} host arrays ({gb:.1f} GB) -> {out_path} ({int(out.bv_bytes) / 1e6:.0f} MB, {int(out.n_unitigs)} unitigs): "
          f"{t1 - t0:.3f} s = {n_inst / (t1 - t0) / 1e9:.2f} Gk-mers/s   (device-resident step {dev_ms:.1f} ms = {n_inst / dev_ms / 1e6:.1f} Gk-mers/s)", flush=True)
    lib.snk_free(C.byref(out))
if ref_units is not None:
    off, bases = graphio.read_bv(out_path)
    print("unitig file == device path (BVComp order):", graphio.arrays_to_unitigs(off, bases) == ref_units, flush=True)
lib.snk_ctx_destroy(h)
