"""round 6: the DF seam's slab upload as one contiguous copy against five copies per slab, on the same box (tuning build -DSNK_DF_MULTI_COPY).
usage: python tools/r6_df_copy_ab.py [reads]"""
import os, subprocess, sys, tempfile, shutil, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from supernova_amd import dfin
    from supernova_amd.engine import Engine, Params
    e = Engine(0)
    e.reserve(int(130e9))
    with dfin.DfFiles(sys.argv[3]) as f:
        ws = []
        for rep in range(5):
            t0 = time.perf_counter()
            res, st = f.count_graph(e, Params(K=48, sorted_table=False), read_len=150)
            res.bv_image()
            ws.append(round(time.perf_counter() - t0, 3))
            del res
        print(os.environ.get("SNK_LIB_PATH", "default build"), "walls", ws, "io wait", round(st["io_wait_seconds"], 3), flush=True)
    sys.exit(0)
from supernova_amd import dfin, synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
td = Path(tempfile.mkdtemp(prefix="snk_dfab_", dir="/tmp"))
try:
    sp = synth.synth_params(n, seed=0x5EED0AB0, unbarcoded_ppm=0)
    dfin.write_synth_df(td / "reads", sp, qual_jitter=8)
    subprocess.run(["bash", str(ROOT / "tools" / "build_variant.sh"), "dfmulti", "-DSNK_DF_MULTI_COPY"], env=dict(os.environ, FILES="snk_dfin"), check=True, capture_output=True)
    for rnd in range(2):
        for lib in ("", str(ROOT / "supernova_amd" / "variants" / "libsnk_dfmulti.so")):
            env = dict(os.environ)
            if lib:
                env["SNK_LIB_PATH"] = lib
            subprocess.run([sys.executable, __file__, str(n), "child", str(td / "reads")], env=env, check=True)
finally:
    shutil.rmtree(td, ignore_errors=True)
