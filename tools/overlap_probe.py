"""Do two steps that run at the same time on ONE GPU (two contexts, two streams, two host threads) finish sooner than one after the other?
The partition kernel stands at the device's atomic rate, the count kernel at VALU issue: if the hardware co-schedules workgroups of both,
a step that pipelines bucket ranges (partition of range r + 1 next to the count of range r) would gain what this probe shows.
usage: python tools/overlap_probe.py reads_per_engine [reps]"""
import sys, threading, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params

per = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.cuda.set_device(0)
sp = synth.synth_params(2 * per, seed=0x5EED0002)
engs, data, streams = [], [], []
for r in range(2):
    e = Engine(0)
    engs.append(e)
    data.append(e.synth(sp, first=r * per, n=per))
    streams.append(torch.cuda.Stream())
torch.cuda.synchronize()
P = Params(K=48, sorted_table=False)


def step(r):
    rows, quals, bc = data[r]
    return engs[r].count_graph(rows, sp.read_len, quals=quals, bc=bc, params=P)


for r in range(2):          # arena + bucket-size history
    for _ in range(2):
        res = step(r)
torch.cuda.synchronize()
print(f"one engine, {per} reads: phases {res.phase_ms}", flush=True)

t0 = time.perf_counter()
for _ in range(reps):
    step(0); step(1)
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / reps * 1e3
print(f"sequential: {seq:.1f} ms per pair of steps", flush=True)

for stagger_ms in (0.0, 15.0, 30.0):
    bar = threading.Barrier(2)
    walls = [0.0, 0.0]

    def worker(r):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[r]):
            bar.wait()
            if r == 1 and stagger_ms:
                time.sleep(stagger_ms * 1e-3)
            t = time.perf_counter()
            for _ in range(reps):
                step(r)
            streams[r].synchronize()
            walls[r] = time.perf_counter() - t

    t0 = time.perf_counter()
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    print(f"concurrent (second thread starts {stagger_ms:.0f} ms later): {wall:.1f} ms per pair of steps = {wall / seq:.3f} of sequential "
          f"(threads: {walls[0] / reps * 1e3:.1f} / {walls[1] / reps * 1e3:.1f} ms per step)", flush=True)
