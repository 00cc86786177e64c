import sys, time, ctypes as C, os
sys.path.insert(0,'/root/repo')
from supernova_amd import ingest, synth, lib as _lib
from pathlib import Path
import tempfile
nf=int(sys.argv[1]) if len(sys.argv)>1 else 8
thr=int(sys.argv[2]) if len(sys.argv)>2 else nf
td=Path(tempfile.mkdtemp())
sp=synth.synth_params(2*nf*100000, seed=5)
paths,text=ingest.write_synth_fasth(td, sp, nf, 100000, workers=8)
L=_lib.load()
class B(C.Structure):
    _fields_=[("n_pairs",C.c_uint64),("file",C.c_uint32),("max_len",C.c_uint32),("first_pair",C.c_uint64),("ascii",C.c_void_p),("quals",C.c_void_p),("lens",C.c_void_p),("bc_fields",C.c_void_p),("text_bytes",C.c_uint64),("token",C.c_uint64)]
arr=(C.c_char_p*nf)(*[p.encode() for p in paths])
err=C.create_string_buffer(512)
for rep in range(2):
    s=C.c_void_p()
    t0=time.perf_counter()
    L.snk_fasth_open.argtypes=[C.POINTER(C.c_char_p),C.c_uint32,C.c_uint32,C.c_uint32,C.c_uint32,C.c_uint32,C.POINTER(C.c_void_p),C.c_char_p,C.c_size_t]
    rc=L.snk_fasth_open(arr,nf,160,32768,thr,0,C.byref(s),err,512)
    assert rc==0, err.value
    n=0
    while True:
        b=B()
        L.snk_fasth_next.argtypes=[C.c_void_p,C.c_void_p,C.c_char_p,C.c_size_t]
        rc=L.snk_fasth_next(s,C.byref(b),err,512); assert rc==0, err.value
        if b.n_pairs==0: break
        n+=b.n_pairs
        L.snk_fasth_release.argtypes=[C.c_void_p,C.c_void_p]
        L.snk_fasth_release(s,C.byref(b))
    t1=time.perf_counter()
    L.snk_fasth_close.argtypes=[C.c_void_p]; L.snk_fasth_close(s)
    print(f"{nf} files, {thr} threads: {t1-t0:.3f} s, {text/1e9/(t1-t0):.3f} GB/s text, {text/1e9/(t1-t0)/thr:.3f} per thread, pairs {n}")
