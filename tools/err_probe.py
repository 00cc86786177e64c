"""The step on the bench's data and on the two error-rich models (bench.py's robust rows), phases and bucket counts: what a tuning
build of the count kernel (tools/build_variant.sh, SNK_LIB_PATH=...) does off the operating point.
usage: python tools/err_probe.py [n_reads] [rows: comma list of headline,e06,e15,cov28,crowded,human] [count]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
want = sys.argv[2].split(",") if len(sys.argv) > 2 else ["headline", "e06", "e15"]
graph = not (len(sys.argv) > 3 and sys.argv[3] == "count")      # "count": stop after the table (a variant whose table outgrows the graph stage's chunks)
MODELS = {"headline": {}, "e06": dict(sub_ppm=6000), "e15": dict(sub_ppm=15000, lowq_tail_ppm=500000), "e10": dict(sub_ppm=10000), "e15n": dict(sub_ppm=15000), "e25": dict(sub_ppm=25000), "cov28": dict(genome_len=n * 150 // 28),
          "crowded": dict(repeat_mode=16),      # the bench's reads over a genome with ~1 site per canonical 16-mer value (human: 1.4), at 56x
          "human": dict(genome_len=3_100_000_000)}      # a genome of human size under these reads (4.8x at 1e8 reads: what the minimiser space of configs 3-5 looks like)
eng = Engine(0)
for name in want:
    sp = synth.synth_params(n, seed=0x5EED0042, **MODELS[name])
    rows, quals, bc = eng.synth(sp)
    torch.cuda.synchronize()
    for rep in range(4):
        t0 = time.perf_counter()
        r = eng.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False, graph=graph))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if rep in (0, 1, 3):
            print(f"{name} call {rep}: {dt:.1f} ms | " + " ".join(f"{k} {v:.1f}" for k, v in r.phase_ms.items() if k in ("partition", "count", "graph")) +
                  f" | supermers {r.n_supermers} repartitioned {r.repartitioned} buckets {r.n_buckets} split {r.buckets_split} kmers {r.n_kmers} unitigs {r.n_unitigs} max_slots {r.max_slots_used}", flush=True)
    del rows, quals, bc, r
