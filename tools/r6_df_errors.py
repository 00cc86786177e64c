"""round 6: the DF seam on error-rich reads.  A streamed job cannot look at its first buckets and partition again (its slabs are gone), so a
one-shot process (snk_mspedges) meets such data without any history: how much slower is that first call than a call that knows the data?
usage: python tools/r6_df_errors.py [reads=1e8] [sub_ppm,...]"""
import sys, time, tempfile, shutil
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from supernova_amd import dfin, synth
from supernova_amd.engine import Engine, Params

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
rates = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2000, 6000, 15000]
for ppm in rates:
    td = Path(tempfile.mkdtemp(prefix="snk_dferr_", dir="/tmp"))
    try:
        sp = synth.synth_params(n, seed=0x5EED0E44, unbarcoded_ppm=0, sub_ppm=ppm)
        dfin.write_synth_df(td / "reads", sp, qual_jitter=8)
        for label, mode in (("forced streamed (what round 6 did first)", 2), ("the library's choice", None)):
            e = Engine(0)
            e.reserve(int(130e9))
            if mode is not None:
                e.set_option("df_stream", mode)
            with dfin.DfFiles(td / "reads") as f:
                print(f"sub_ppm {ppm}: {label}, fresh context: calls (wall s, mode, phases ms, buckets, split, table slots usable):", flush=True)
                for rep in range(3):
                    t0 = time.perf_counter()
                    res, st = f.count_graph(e, Params(K=48, sorted_table=False), read_len=150)
                    wall = time.perf_counter() - t0
                    print("   ", (round(wall, 3), st["mode"], {k: round(v, 1) for k, v in res.phase_ms.items() if k in ("partition", "count", "graph")}, int(res.n_buckets),
                                  int(res.buckets_split), e.get_tuning()["last_count_limit"]), flush=True)
                    del res
            e.close()
    finally:
        shutil.rmtree(td, ignore_errors=True)
