"""round 6: the WHOLE job of BASELINE config 3 (1.2 B x 150 bp synthetic linked reads over a 3.2 Gb genome, k = 48) on ONE MI355X.
The reads are generated slab by slab, trimmed, and kept in the compact form the DF seam keeps (packed rows + good lengths + barcode ids,
46 bytes per read: 55 GB); the resident step (snk_dev_count_graph) then runs on them in as many bucket-range passes as its memory plan
asks for.  Properties checked on the device: every count >= min_freq, the spectrum and the unitig lengths add up to the table size,
and the second call returns the first one's table checksum and unitigs.
usage: python tools/r6_full_job.py [reads=1.2e9] [slab=5e7] [calls=2] [minimiser=auto|16|20] [debug]      env: GENOME_LEN=, K=60, GROUPED=1 (per-barcode graphs), SUB_PPM= / LOWQ_TAIL_PPM= (the read model's errors)"""
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from supernova_amd import synth  # noqa: E402
from supernova_amd.engine import Engine, Params  # noqa: E402


class _DevArr:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 3}


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_200_000_000
    slab = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    mini = sys.argv[4] if len(sys.argv) > 4 else "auto"
    e = Engine(0)
    dev = torch.device("cuda", 0)
    ov = {"genome_len": int(float(os.environ["GENOME_LEN"]))} if os.environ.get("GENOME_LEN") else {}
    if os.environ.get("SUB_PPM"):
        ov.update(sub_ppm=int(os.environ["SUB_PPM"]), lowq_tail_ppm=int(os.environ.get("LOWQ_TAIL_PPM", "0")))
    sp = synth.synth_params(n, seed=0x5EED0C30, **ov)
    long_min = mini == "20" or (mini == "auto" and int(sp.genome_len) >= 1_500_000_000)
    t0 = time.perf_counter()
    rows = torch.empty((n, 10), dtype=torch.int32, device=dev)
    gl = torch.empty((n,), dtype=torch.int16, device=dev)
    bc = torch.empty((n,), dtype=torch.int32, device=dev)
    for first in range(0, n, slab):
        m = min(slab, n - first)
        r, q, b = e.synth(sp, first, m)
        g = e.trim(q, 150, K=int(os.environ.get("K", "48")))
        rows[first:first + m] = r
        gl[first:first + m] = g
        bc[first:first + m] = b
        del r, q, b, g
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    print(f"reads {n} genome {int(sp.genome_len)} compact form {(rows.nbytes + gl.nbytes + bc.nbytes) / 2**30:.1f} GiB made in {time.perf_counter() - t0:.1f} s; "
          f"device free {free / 2**30:.1f} of {total / 2**30:.1f} GiB; minimisers of {20 if long_min else 16}", flush=True)
    K = int(os.environ.get("K", "48"))
    grouped = os.environ.get("GROUPED") == "1"
    params = Params(K=K, sorted_table=False, grouped=True, min_bc=0) if grouped else Params(K=K, sorted_table=False, long_minimiser=long_min)
    seen = None
    out = []
    for call in range(calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = e.count_graph(rows, 150, good_len=gl, bc=None, group=bc, params=params) if grouped else e.count_graph(rows, 150, good_len=gl, bc=bc, params=params)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        nk, nu = res.n_kmers, res.n_unitigs
        # properties, in pieces (the table is tens of GB)
        chk, cmin, piece = 0, 1 << 30, 32_000_000
        for a in range(0, nk, piece):
            c = min(piece, nk - a)
            keys = torch.as_tensor(_DevArr(res.raw.keys + 16 * a, 2 * c, "<i8"), device=dev).view(c, 2)
            cnt = torch.as_tensor(_DevArr(res.raw.counts + 4 * a, c, "<i4"), device=dev)
            ctx = torch.as_tensor(_DevArr(res.raw.ctx + a, c, "|u1"), device=dev)
            cmin = min(cmin, int(cnt.min()))
            chk = (chk + int((keys[:, 1] * 0x9E3779B97F4A7C15 + keys[:, 0] * 0x42B2AE3D27D4EB4F + cnt.to(torch.int64) * 0x165667B19E3779F9
                              + ctx.to(torch.int64) * 0x27D4EB2F165667C5).sum())) & ((1 << 64) - 1)
            del keys, cnt, ctx
        spec = torch.as_tensor(_DevArr(res.raw.spectrum, int(res.raw.spectrum_bins), "<i8"), device=dev)
        off = torch.as_tensor(_DevArr(res.raw.unitig_off, nu + 1, "<i8"), device=dev)
        ok_spec = int(spec.sum()) == nk
        ok_len = int((off[1:] - off[:-1] - (K - 1)).sum()) == nk
        ubytes = res.unitig_total_bases
        uchk = 0
        for a in range(0, ubytes, 1 << 26):
            c = min(1 << 26, ubytes - a)
            ub = torch.as_tensor(_DevArr(res.raw.unitig_bases + a, c, "|u1"), device=dev)
            uchk = (uchk * 1000003 + int((ub.to(torch.int64) * (torch.arange(c, device=dev, dtype=torch.int64) % 1000033 + 1)).sum())) & ((1 << 64) - 1)
            del ub
        sig = (res.n_instances, nk, chk, nu, int(off.sum()), uchk)
        row = dict(call=call, wall_s=round(wall, 3), Gkmers_per_s=round(res.n_instances / wall / 1e9, 2), instances=int(res.n_instances), retained_kmers=int(nk), unitigs=int(nu),
                   passes=e.last_partition_passes(), count_limit=e.last_count_limit(), buckets=int(res.n_buckets), buckets_split=int(res.buckets_split),
                   scratch_gb=round(res.scratch_bytes / 2**30, 1), repartitioned=int(res.repartitioned), n_boundary=res.n_boundary, n_fragments=res.n_fragments, n_circles=res.n_circles, rank_rounds=res.rank_rounds, overflow_supermers=int(getattr(res, "n_overflow", 0)),
                   phase_ms={k: round(v, 1) for k, v in res.phase_ms.items()}, min_count=cmin, table_checksum=hex(chk), spectrum_adds_up=ok_spec, unitig_lengths_add_up=ok_len,
                   same_as_first_call=(seen is None or sig == seen))
        if len(sys.argv) > 5 and sys.argv[5] == "debug":
            import numpy as np
            h_off = off.cpu().numpy()
            h_b = np.empty(ubytes, dtype=np.uint8)
            for a in range(0, ubytes, 1 << 28):
                c = min(1 << 28, ubytes - a)
                h_b[a:a + c] = torch.as_tensor(_DevArr(res.raw.unitig_bases + a, c, "|u1"), device=dev).cpu().numpy()
            def keys_of(o, b):
                fw = [bytes(b[o[u]:o[u] + 48]) for u in range(len(o) - 1)]
                rv = [bytes((3 - b[o[u + 1] - 48:o[u + 1]])[::-1]) for u in range(len(o) - 1)]
                return fw, rv
            fw, rv = keys_of(h_off, h_b)
            print("debug: call", call, "first-k-mer keys ascending:", all(fw[i] < fw[i + 1] for i in range(len(fw) - 1)), "unitigs whose reverse complement starts smaller:",
                  sum(1 for x, y in zip(fw, rv) if y < x), flush=True)
            if call == 0:
                first_off, first_b = h_off, h_b
                first_fw, first_rv = set(fw), set(rv)
                first_len = {k: int(h_off[u + 1] - h_off[u]) for u, k in enumerate(fw)}
            else:
                print("debug: first k-mers also first k-mers of call 0:", sum(1 for x in fw if x in first_fw), "that are call 0's reverse ends:", sum(1 for x in fw if x in first_rv and x not in first_fw),
                      "neither:", sum(1 for x in fw if x not in first_fw and x not in first_rv), flush=True)
                same_len = sum(1 for u, k in enumerate(fw) if first_len.get(k) == int(h_off[u + 1] - h_off[u]))
                print("debug: same first k-mer and same length:", same_len, flush=True)
                ln0, ln1 = np.diff(first_off), np.diff(h_off)
                d = np.nonzero(ln0 != ln1)[0]
                print("debug: unitigs whose length differs", len(d), "first", d[:5], "lengths", ln0[d[:5]], ln1[d[:5]], flush=True)
                db = np.nonzero(first_b != h_b)[0] if len(first_b) == len(h_b) else []
                print("debug: bases that differ", len(db), "first at", db[:3] if len(db) else None, "in unitig", (np.searchsorted(first_off, db[:3], side="right") - 1) if len(db) else None, flush=True)
                if len(db):
                    u = int(np.searchsorted(first_off, db[0], side="right") - 1)
                    print("debug: unitig", u, "len", ln0[u], ln1[u], "offset in unitig", int(db[0] - first_off[u]), "last diff at", int(db[-1]), "unitig", int(np.searchsorted(first_off, db[-1], side="right") - 1), flush=True)
                    # multiset of lengths equal?
                    print("debug: sorted lengths equal", bool((np.sort(ln0) == np.sort(ln1)).all()), flush=True)
        if seen is not None and sig != seen:
            row["differs_in"] = [nm for nm, x, y in zip(("instances", "retained_kmers", "table_checksum", "unitigs", "unitig_offsets", "unitig_bases"), sig, seen) if x != y]
        seen = seen or sig
        print(json.dumps(row), flush=True)
        out.append(row)
        assert cmin >= 3 and ok_spec and ok_len
        del res, spec, off
    assert all(r["same_as_first_call"] for r in out)
    return out


if __name__ == "__main__":
    main()
