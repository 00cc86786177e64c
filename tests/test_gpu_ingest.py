"""f3 on the device: FASTH files -> HBM through snk_dev_ingest_fasth == the synthetic model's reads generated in place, and the
count+graph of both is the same."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ingest_fasth_matches_model_and_counts_alike(snk, tmp_path):
    import torch
    from supernova_amd import ingest, synth
    from supernova_amd.engine import Engine, Params
    n_files, ppf = 9, 11_111
    n = 2 * n_files * ppf
    sp = synth.synth_params(n, seed=0x5EED0A11)
    paths, text = ingest.write_synth_fasth(tmp_path, sp, n_files, ppf)
    rows, quals, bc = synth.synth_host(sp)
    e = Engine(0)
    wl = ingest.synth_whitelist(int(bc.max()) + 5)
    for threads, bp in ((0, 0), (2, 1000)):
        dr = ingest.ingest_fasth(e, paths, sp.read_len, wl, threads=threads, batch_pairs=bp)
        assert dr.n_reads == n and dr.stats["text_bytes"] == text and dr.stats["max_len"] == sp.read_len
        dl = lambda ptr, shape, dt: (lambda a: (e._download(ptr, a.ctypes.data, a.nbytes), a)[1])(np.empty(shape, dtype=dt))
        got_rows = dl(dr.raw.rows, (n, int(dr.raw.row_words)), np.uint32)
        got_q = dl(dr.raw.quals, (n, int(dr.raw.qstride)), np.uint8)
        got_l = dl(dr.raw.lens, (n,), np.uint16)
        got_bc = dl(dr.raw.bc, (n,), np.int32)
        assert np.array_equal(got_rows, rows) and np.array_equal(got_q[:, :sp.read_len], quals[:, :sp.read_len])
        assert np.all(got_q[:, sp.read_len:] == 0) and np.all(got_l == sp.read_len) and np.array_equal(got_bc, bc)
        res = e.count_graph_reads(dr.dev_reads(), Params(K=48))
        u1, k1, c1 = res.unitigs(), res.keys(), res.counts()
        dev = torch.device("cuda", 0)
        ref = e.count_graph(torch.from_numpy(rows.view(np.int32)).to(dev), sp.read_len, quals=torch.from_numpy(np.ascontiguousarray(quals)).to(dev),
                            bc=torch.from_numpy(bc).to(dev), params=Params(K=48))
        assert u1 == ref.unitigs() and np.array_equal(k1, ref.keys()) and np.array_equal(c1, ref.counts())
        dr.close()
    e.close()


def test_streamed_ingest_equals_the_resident_path(snk, tmp_path):
    """snk_dev_ingest_count_graph: the decoded batches go straight into a streamed job (snk_dev_stream_*: partitioned while the next ones
    are inflated, the reads never resident as a whole) -- table, counts, contexts, spectrum and unitigs equal those of the reads
    generated in place and counted in one resident call; small batches so that a job has dozens of slabs in any arrival order; with the
    host's libdeflate and with zlib's streaming inflate."""
    import os
    import torch
    from supernova_amd import ingest, synth
    from supernova_amd.engine import Engine, Params
    n_files, ppf = 7, 9_973
    n = 2 * n_files * ppf
    sp = synth.synth_params(n, seed=0x5EED0A12)
    paths, text = ingest.write_synth_fasth(tmp_path, sp, n_files, ppf)
    rows, quals, bc = synth.synth_host(sp)
    e = Engine(0)
    wl = ingest.synth_whitelist(int(bc.max()) + 5)
    dev = torch.device("cuda", 0)
    ref = e.count_graph(torch.from_numpy(rows.view(np.int32)).to(dev), sp.read_len, quals=torch.from_numpy(np.ascontiguousarray(quals)).to(dev),
                        bc=torch.from_numpy(bc).to(dev), params=Params(K=48))
    ur, kr, cr, xr, sr, glr = ref.unitigs(), ref.keys(), ref.counts(), ref.ctx(), ref.spectrum(), sorted(ref.good_len().tolist())
    for threads, bp, hint in ((0, 0, 0), (3, 1500, n), (1, 700, 0)):
        res, st = ingest.ingest_count_graph(e, paths, sp.read_len, wl, params=Params(K=48), threads=threads, batch_pairs=bp, total_reads_hint=hint)
        assert st["n_reads"] == n and st["text_bytes"] == text and res.n_reads == n
        assert res.unitigs() == ur and np.array_equal(res.keys(), kr) and np.array_equal(res.counts(), cr) and np.array_equal(res.ctx(), xr)
        assert np.array_equal(res.spectrum(), sr)
        assert sorted(res.good_len().tolist()) == glr          # (arrival order)
    # a bound that is too small is refused, not overrun
    from supernova_amd.lib import SnkError
    with pytest.raises(SnkError, match="upper bound"):
        ingest.ingest_count_graph(e, paths, sp.read_len, wl, params=Params(K=48), total_reads_hint=n // 3)
    e.close()
