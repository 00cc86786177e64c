# round 5: count-kernel geometry (threads x table slots per workgroup) at matching bucket sizes; whole step, 100 M reads
cat > /tmp/g.py <<'PY'
import os, sys
sys.path.insert(0, '/root/repo')
import torch
from supernova_amd import synth
from supernova_amd.engine import Engine, Params
e = Engine(0); sp = synth.synth_params(100_000_000, seed=0x5EED0001); rows, quals, bc = e.synth(sp)
for _ in range(3):
    r = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=48, sorted_table=False))
print(os.environ.get("SNK_LIB_PATH", "default").split("/")[-1], "target", os.environ.get("SNK_TARGET_INST", "-"), "buckets", r.n_buckets, "split", r.buckets_split,
      "partition %.1f count %.1f graph %.1f total %.1f" % (r.phase_ms["partition"], r.phase_ms["count"], r.phase_ms["graph"], r.phase_ms["total"]), flush=True)
PY
python /tmp/g.py 2>&1 | grep -v amdgpu
for t in 2000 2600 3200; do SNK_TARGET_INST=$t SNK_LIB_PATH=supernova_amd/variants/libsnk_t384s1024.so python /tmp/g.py 2>&1 | grep -v amdgpu; done
for t in 1600 2000 2400; do SNK_TARGET_INST=$t SNK_LIB_PATH=supernova_amd/variants/libsnk_t512s1024.so python /tmp/g.py 2>&1 | grep -v amdgpu; done
for t in 2600 3200 3800; do SNK_TARGET_INST=$t SNK_LIB_PATH=supernova_amd/variants/libsnk_t256s1024.so python /tmp/g.py 2>&1 | grep -v amdgpu; done
