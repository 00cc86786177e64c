"""round 6: what ONE run of the stage's C++ host costs end to end -- snk_mspedges LR=reads.fastb OUT=asm_graph.bv in a fresh process (context, arena,
ingest of the three stage-input files, count + graph, the .bv file) -- page cache warm.  usage: python tools/r6_oneshot.py [reads=1e8] [runs=3] [extra KEY=VALUE ...]"""
import os, shutil, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from supernova_amd import dfin, synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = sys.argv[3:]
td = Path(tempfile.mkdtemp(prefix="snk_one_", dir="/tmp"))
try:
    sp = synth.synth_params(n, seed=0x5EED0AB0, unbarcoded_ppm=0)
    dfin.write_synth_df(td / "reads", sp, qual_jitter=8)
    exe = ROOT / "supernova_amd" / "bin" / "snk_mspedges"
    for r in range(runs):
        t0 = time.perf_counter()
        pr = subprocess.run([str(exe), f"LR={td / 'reads.fastb'}", f"OUT={td / 'asm_graph.bv'}", "READ_LEN=150"] + extra, capture_output=True, text=True, env=dict(os.environ))
        wall = time.perf_counter() - t0
        print(f"run {r}: rc {pr.returncode} wall {wall:.3f} s | " + " | ".join(l.strip()[:400] for l in pr.stderr.splitlines() if "snk_mspedges" in l or "arena" in l), flush=True)
finally:
    shutil.rmtree(td, ignore_errors=True)
