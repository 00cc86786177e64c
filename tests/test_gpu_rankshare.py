"""One rank's share of BASELINE configs 3, 4 and 5 (1.2 B x 150 bp on 8 GPUs = 150 M reads per GPU) on ONE MI355X:

  C3  K=48, minimiser-sharded path (ShardedEngine over a real one-rank RCCL group: histogram / record / query / link
      exchanges, ranged count launches, fragment gather, join)
  C4  the same at K=60
  C5  per-barcode local graphs (SNK_F_GROUPED; replicas only, no collective)

An 8-GPU node is not available to the test suite, so what is checked is everything that does not need a second device:
the per-rank kernels and buffers at their production size (150 M reads, 15.3 G k-mer instances, ~33 GB of supermer
records through the exchange), with size-independent properties -- strictly ascending keys, every count >= min_freq,
spectrum and unitig lengths adding up to the table -- and the checksum of checksums: the sharded path and the one-GPU
path (different bucket counts, segment layouts, graph stage entry points) must give the same table and the same
unitigs, byte for byte, on the same 150 M reads.  Parity against the reference itself is pinned at 2 M / 10 M reads
(test_gpu_bigparity.py) and on the golden cases."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_SHARE = 150_000_000


class _DevArr:
    """A device array of the library seen by torch (no copy): __cuda_array_interface__ over the raw pointer."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _view(ptr, n, typestr):
    import torch
    if n == 0:
        return torch.zeros(0, dtype={"<i8": torch.int64, "<i4": torch.int32, "|u1": torch.uint8}[typestr], device="cuda")
    return torch.as_tensor(_DevArr(ptr, n, typestr), device="cuda")


def _table_props(keys_ptr, counts_ptr, ctx_ptr, nk, min_freq, sorted_keys):
    """On-device checks of a retained table + an order-independent 64-bit checksum of (key, count, context)."""
    import torch
    keys = _view(keys_ptr, 2 * nk, "<i8").view(nk, 2)      # lo, hi
    cnt = _view(counts_ptr, nk, "<i4")
    ctx = _view(ctx_ptr, nk, "|u1")
    lo, hi = keys[:, 0], keys[:, 1]
    if sorted_keys:
        f = lambda t: t ^ torch.tensor(-(1 << 63), dtype=torch.int64, device="cuda")
        h0, h1, l0, l1 = f(hi[:-1]), f(hi[1:]), f(lo[:-1]), f(lo[1:])
        assert bool(((h1 > h0) | ((h1 == h0) & (l1 > l0))).all())
    assert int(cnt.min()) >= min_freq
    chk = (hi * 0x9E3779B97F4A7C15 + lo * 0x42B2AE3D27D4EB4F + cnt.to(torch.int64) * 0x165667B19E3779F9
           + ctx.to(torch.int64) * 0x27D4EB2F165667C5).sum()
    return int(chk)


@pytest.fixture(scope="module")
def rccl_group(snk):
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("K", [48, 60])
def test_rank_share_sharded_equals_single(rccl_group, K):
    """C3 / C4: 150 M reads through the sharded path (real RCCL group of one rank) == the one-GPU path, at full per-rank size."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine
    n = N_SHARE
    e = Engine(0)
    try:
        # 150 M reads at the job's 56x coverage (a 402 Mb genome): the read, record and k-mer volumes one rank of the 8-GPU job
        # handles (its slab of the 1.2 B reads alone would be 7x coverage of the 3.2 Gb genome -- 2.5 G retained k-mers,
        # nothing like a rank's load after the exchange)
        sp = synth.synth_params(n, seed=0x5EED0003 if K == 48 else 0x5EED0004)
        rows, quals, bc = e.synth(sp)
        r = e.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=K))
        nk = r.n_kmers
        assert r.n_instances > 0.9 * n * (150 - K + 1) and nk > 0
        chk1 = _table_props(r.raw.keys, r.raw.counts, r.raw.ctx, nk, 3, sorted_keys=True)
        spec = _view(r.raw.spectrum, int(r.raw.spectrum_bins), "<i8")
        assert int(spec.sum()) == nk
        off1 = _view(r.raw.unitig_off, r.n_unitigs + 1, "<i8").clone()
        bases1 = _view(r.raw.unitig_bases, r.unitig_total_bases, "|u1").clone()
        assert int((off1[1:] - off1[:-1] - (K - 1)).sum()) == nk
        single = dict(n_inst=r.n_instances, nk=nk, nu=r.n_unitigs)
        del r
        e.close()
        torch.cuda.empty_cache()

        e = Engine(0)
        sh = ShardedEngine(e, rccl_group)
        assert sh.kind == "rccl" and sh.world == 1
        s = sh.count_graph(rows, 150, quals=quals, bc=bc, params=Params(K=K))
        assert (s.n_instances, s.n_kmers) == (single["n_inst"], single["nk"])
        chk2 = _table_props(s.raw.keys, s.raw.counts, s.raw.ctx, s.n_kmers, 3, sorted_keys=False)
        assert chk2 == chk1
        u = s.raw
        assert int(u.n_unitigs) == single["nu"]
        off2 = _view(u.unitig_off, int(u.n_unitigs) + 1, "<i8")
        bases2 = _view(u.unitig_bases, int(u.unitig_total_bases), "|u1")
        assert torch.equal(off1, off2) and torch.equal(bases1, bases2)
    finally:
        e.close()
        torch.cuda.empty_cache()


def test_rank_share_grouped(snk):
    """C5: 150 M reads as per-barcode groups (187 500 barcodes of 800 reads) in one grouped run; properties + independence
    of the bucket count."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params
    n = N_SHARE
    e = Engine(0)
    try:
        sp = synth.synth_params(n, seed=0x5EED0005)
        assert n % (2 * sp.pairs_per_bc) == 0
        rows, quals, bc = e.synth(sp)
        # per-barcode coverage is far below 1x: with the production filter (3 observations inside ONE barcode) next to
        # nothing survives, so the share is also run with min_freq 1 on a tenth of it (every k-mer of every group retained)
        outs = []
        for nb in (0, 3_000_017):
            r = e.count_graph(rows, 150, quals=quals, bc=None, group=bc, params=Params(K=48, grouped=True, min_bc=0, sorted_table=False,
                                                                                      n_buckets=nb))
            nk = r.n_kmers
            chk = _table_props(r.raw.keys, r.raw.counts, r.raw.ctx, nk, 3, sorted_keys=False) if nk else 0
            spec = _view(r.raw.spectrum, int(r.raw.spectrum_bins), "<i8")
            assert int(spec.sum()) == nk
            off = _view(r.raw.unitig_off, r.n_unitigs + 1, "<i8")
            assert int((off[1:] - off[:-1] - 47).sum()) == nk
            outs.append((r.n_instances, nk, chk, r.n_unitigs))
        assert outs[0] == outs[1] and outs[0][0] > 14_000_000_000
        m = n // 10
        outs = []
        del r
        e.release_cache()          # the scratch of the 150 M-read runs goes back to the device: the checks below are torch's, on the same GPU
        for nb in (0, 2_000_003):
            r = e.count_graph(rows[:m], 150, quals=quals[:m], bc=None, group=bc[:m],
                              params=Params(K=48, grouped=True, min_bc=0, min_freq=1, sorted_table=False, n_buckets=nb))
            nk = r.n_kmers
            chk = _table_props(r.raw.keys, r.raw.counts, r.raw.ctx, nk, 1, sorted_keys=False)
            off = _view(r.raw.unitig_off, r.n_unitigs + 1, "<i8")
            assert int((off[1:] - off[:-1] - 47).sum()) == nk
            grp = _view(r.raw.unitig_group, r.n_unitigs, "<i4")
            assert bool((grp[1:] >= grp[:-1]).all())                   # output is group-major
            outs.append((r.n_instances, nk, chk, r.n_unitigs))
        assert outs[0] == outs[1] and outs[0][1] > 1_000_000_000
    finally:
        e.close()
        torch.cuda.empty_cache()
