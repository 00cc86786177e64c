"""Parity above the size of committed dumps: the HIP path against SHA-256 fixtures of the REFERENCE ITSELF
(tests/golden/big_hashes.json, made by tests/golden/make_big_hashes.py from oracle/_ref/snref_driver[60] in the build
container) on BASELINE config 1 (10 M x 150 bp, seed 0x5EED0001) and a 2 M-read instance of the same
model -- one GPU, and 8 simulated ranks of the minimiser-sharded path.  Equal digests = bit-equal good lengths, retained
table (keys, counts, pruned contexts), spectrum and canonical unitigs (tests/bighash.py).

The device generator is the host generator bit for bit (test_gpu_parity.py::test_synth_vs_oracle), so the reads are made
in HBM directly."""
import json
import threading
from pathlib import Path

import numpy as np
import pytest

import bighash

pytestmark = pytest.mark.gpu

FIX = Path(__file__).resolve().parent / "golden" / "big_hashes.json"
HASHES = json.loads(FIX.read_text()) if FIX.exists() else {}
FIELDS = ("n_reads", "n_kmers", "n_unitigs", "unitig_bases", "goodlens", "keys", "counts", "ctx", "hist", "unitigs")


def _case(name):
    if name not in HASHES:
        pytest.skip(f"{name}: no fixture in tests/golden/big_hashes.json (made in the build container)")
    return HASHES[name]


def _compare(dg, exp, fields=FIELDS):
    bad = [f for f in fields if dg[f] != exp[f]]
    assert not bad, {f: (dg[f], exp[f]) for f in bad}


# the count-kernel instantiations the library picks by itself on error-rich data, forced (tests/test_gpu_parity.py COUNT_VARIANTS)
VARIANTS = {"default": ({}, None), "screen": ({"SNK_COUNT_SCREEN_NG": "2"}, 960), "tight": ({"SNK_COUNT_TIGHT": "1920", "SNK_COUNT_SCREEN_NG": "0"}, 1920)}
ROBUST_200K = ["robust_err06_200k", "robust_err15_200k", "robust_cov28_200k", "robust_repeats_200k"]


@pytest.mark.parametrize("name,variant", [(n, "default") for n in ["c1_2m", "c1_10m", "c1_2m_k60", "c1_10m_k60", *ROBUST_200K, "robust_repeats_1m", "robust_repeats_200k_k60"]]
                         + [(n, v) for n in ROBUST_200K + ["c1_2m"] for v in ("screen", "tight")])
def test_one_gpu_vs_reference_digest(snk, name, variant, monkeypatch, tune):
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params
    exp = _case(name)
    K = exp["K"]
    env, limit = VARIANTS[variant]
    for k, v in env.items():
        tune(k, v)
    e = Engine(0)
    try:
        if name == "robust_repeats_1m":
            # at this size the repeat families are 100 copies deep: with the threshold down, their minimiser buckets take the hot path
            # (snk_hot.hip) the 100 M-read runs of bench.py config.robust take on their own
            tune("SNK_HOT_MIN", "1000")
            tune("SNK_HOT_FACTOR", "1")
        sp = synth.synth_params(exp["n_reads"], seed=exp["seed"], **exp.get("overrides", {}))
        rows, quals, bc = e.synth(sp)
        # the reference's K=60 variant has no barcode rule: run without a barcode vector there
        res = e.count_graph(rows, sp.read_len, quals=quals, bc=bc if K == 48 else None, params=Params(K=K))
        hist = res.spectrum().astype(np.int64)
        dg = bighash.digest(res.good_len().astype(np.uint32), res.keys(), res.counts(), res.ctx(), res.unitigs(), hist,
                            kw=3 if K == 48 else 4)
        if exp.get("hist_from") != "reference json":
            # no spectrum file from the reference's K=60 variant: the fixture's histogram is over the retained counts
            hist = np.bincount(np.minimum(res.counts(), (1 << 24) - 1)).astype(np.int64)
            dg["hist"] = bighash.digest(np.zeros(0), np.zeros((0, 4)), [], [], [], hist)["hist"]
        _compare(dg, exp)
        if limit is not None:
            assert e.last_count_limit() == limit          # the forced kernel is the one that ran
        if name == "robust_repeats_1m":
            assert res.n_hot_buckets > 0
    finally:
        e.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("name,W", [("c1_2m", 8), ("c1_10m", 8), ("c1_2m_k60", 3)])
def test_simulated_ranks_vs_reference_digest(snk, name, W):
    """The minimiser-sharded SPMD code (W ranks as threads on one GPU, tensor copies as the transport): the union of
    the ranks' tables and the joined unitigs against the same reference digests."""
    import torch
    from supernova_amd import synth
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine, SimWorld
    exp = _case(name)
    K, n = exp["K"], exp["n_reads"]
    world = SimWorld(W)
    bounds = [(n * r // W) & ~1 for r in range(W)] + [n]       # mates stay together
    out, errs = [None] * W, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            e = Engine(0)
            lo, hi = bounds[r], bounds[r + 1]
            sp = synth.synth_params(n, seed=exp["seed"])
            rows, quals, bc = e.synth(sp, first=lo, n=hi - lo)
            sh = ShardedEngine(e, world.comm(r))
            res = sh.count_graph(rows, sp.read_len, quals=quals, bc=bc if K == 48 else None, params=Params(K=K),
                                 read_index_base=lo)
            out[r] = dict(keys=res.keys(), counts=res.counts(), ctx=res.ctx(), spectrum=res.spectrum().astype(np.int64),
                          unitigs=res.unitigs())
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)
            world.barrier_obj.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    keys = np.concatenate([o["keys"] for o in out])
    counts = np.concatenate([o["counts"] for o in out])
    ctx = np.concatenate([o["ctx"] for o in out])
    order = np.lexsort((keys[:, 3], keys[:, 2], keys[:, 1], keys[:, 0]))
    keys, counts, ctx = keys[order], counts[order], ctx[order]
    if exp.get("hist_from") == "reference json":
        nb = max(len(o["spectrum"]) for o in out)
        hist = sum(np.pad(o["spectrum"], (0, nb - len(o["spectrum"]))) for o in out)
    else:
        hist = np.bincount(np.minimum(counts, (1 << 24) - 1)).astype(np.int64)
    dg = bighash.digest(np.zeros(n, np.uint32), keys, counts, ctx, [u for o in out for u in o["unitigs"]], hist, kw=3 if K == 48 else 4)
    _compare(dg, exp, fields=tuple(f for f in FIELDS if f != "goodlens"))
    torch.cuda.empty_cache()
