"""b1/b2 at rate: reads.fastb / reads.qualp / reads.bci decoded on the device (snk_dfin.hip) == the host readers of snk_formats.hip byte for byte
(which are pinned to files the reference's own writers made), on the golden triple, on ragged triples of this repo's writer (every value
width, multi-block chains, adversarial block cuts), on a >= 1 M-read triple written by the reference (snref_driver ... formats), in rank
slices -- and the streamed count+graph over the slabs == a resident call on the same reads.
Reference: lib/assembly/src/10X/DF.cc:265-272,345,464-469,595-597; feudal/PQVec.cc:86-200; feudal/FeudalControlBlock.h:27-166."""
from pathlib import Path

import numpy as np
import pytest

import goldens
import refio
from test_formats_df import FMT, _random_triple

pytestmark = pytest.mark.gpu


def _dl(e, ptr, shape, dt):
    a = np.empty(shape, dtype=dt)
    if a.nbytes:
        e._download(ptr, a.ctypes.data, a.nbytes)
    return a


def _device_arrays(e, dr, with_bc=True):
    n = dr.n_reads
    rows = _dl(e, dr.raw.rows, (n, int(dr.raw.row_words)), np.uint32)
    q = _dl(e, dr.raw.quals, (n, int(dr.raw.qstride)), np.uint8)
    lens = _dl(e, dr.raw.lens, (n,), np.uint16)
    bc = _dl(e, dr.raw.bc, (n,), np.int32) if with_bc and dr.raw.bc else None
    return rows, q, lens, bc


def _host_arrays(head, n, qstride):
    from supernova_amd import formats
    rows, lens, mx = formats.read_fastb(str(head) + ".fastb")
    q = formats.read_qualp(str(head) + ".qualp", n, qstride)
    bc, nb = formats.read_bci(str(head) + ".bci", n)
    return rows, q, lens, bc, mx


def _check_equal(e, f, head, first=0, n=None, **kw):
    n_all = f.n_reads
    n = n_all - first if n is None else n
    dr = f.ingest(e, first=first, n=n, **kw)
    rows, q, lens, bc = _device_arrays(e, dr)
    qs = int(dr.raw.qstride)
    hr, hq, hl, hb, mx = _host_arrays(head, n_all, qs)
    rw = int(dr.raw.row_words)
    assert dr.read_len == max(mx, 1) or kw.get("read_len")
    assert np.array_equal(lens, hl[first:first + n])
    assert np.array_equal(rows[:, :hr.shape[1]], hr[first:first + n]) and not rows[:, hr.shape[1]:].any()
    assert np.array_equal(q, hq[first:first + n])           # (rows beyond a read's values are zero on both sides)
    assert np.array_equal(bc, hb[first:first + n])
    assert rw * 16 == qs
    dr.close()
    return n


def test_golden_triple(snk):
    """the files the reference's writers made (tests/golden/formats): device decode == host readers == the reads they were made from"""
    from supernova_amd import dfin
    from supernova_amd.engine import Engine
    e = Engine(0)
    c = goldens.load("synth_2k_err")
    order = np.load(FMT / "order.npy")
    with dfin.DfFiles(FMT / "reads") as f:
        assert f.n_reads == 2000 and f.n_barcodes == int(c.bc.max()) + 1 and f.max_len(e) == 150
        for slab, threads in ((0, 0), (16, 1), (334, 3), (2000, 2)):
            dr = f.ingest(e, slab_reads=slab, threads=threads)
            rows, q, lens, bc = _device_arrays(e, dr)
            assert np.array_equal(rows, c.rows[order]) and np.array_equal(lens, c.lens[order]) and np.array_equal(bc, c.bc[order])
            assert np.array_equal(q[:, :150], c.quals[order]) and not q[:, 150:].any()
            dr.close()
        _check_equal(e, f, FMT / "reads", first=778, n=1001, slab_reads=100)
    # without the index: no barcode array
    with dfin.DfFiles(FMT / "reads", with_bci=False) as f:
        dr = f.ingest(e)
        assert not dr.raw.bc and dr.n_reads == 2000
        dr.close()
    e.close()


@pytest.mark.parametrize("adversarial", [0, 0xBAD5EED])
def test_ragged_triples_every_width(snk, tmp_path, adversarial):
    """ragged lengths (0, 1, 150), rows of every value width 0..7 (the adversarial writer cuts blocks at random and widens values: nBits up
    to 7, chains of up to dozens of blocks), barcodes with empty ordinals; whole file, odd slabs, rank slices that start at odd reads"""
    from supernova_amd import dfin
    from supernova_amd.engine import Engine
    n = 70_001
    rows, lens, q, bc = _random_triple(n, 99 + adversarial)
    bc[bc > 5] += 3                        # ordinals without reads
    head = tmp_path / "reads"
    dfin.write_df(head, rows, q, bc, lens=lens, read_len=150, adversarial=adversarial)
    e = Engine(0)
    with dfin.DfFiles(head) as f:
        assert f.n_reads == n
        _check_equal(e, f, head)
        _check_equal(e, f, head, slab_reads=4099, threads=5)
        for r in range(3):                                  # three "ranks"
            lo, hi = n * r // 3, n * (r + 1) // 3
            _check_equal(e, f, head, first=lo, n=hi - lo, slab_reads=10_000)
        _check_equal(e, f, head, first=n - 1, n=1)
        _check_equal(e, f, head, first=5, n=0)
        _check_equal(e, f, head, read_len=160)              # the caller's row length (a job's ranks agree on one)
        from supernova_amd.lib import SnkError
        with pytest.raises(SnkError, match="bad length"):
            f.ingest(e, read_len=100)                       # rows too short for the reads: refused, not truncated
    e.close()


def test_long_quality_chains_split_the_slab(snk, tmp_path):
    """one-value blocks (3 bytes per value: 450 bytes for a 150-base read) overflow a slot's quality section: the slab is decoded in pieces
    over the same offset tables"""
    from supernova_amd import dfin
    from supernova_amd.engine import Engine
    n = 30_000
    rng = np.random.default_rng(5)
    rows = rng.integers(0, 1 << 32, (n, 10), dtype=np.uint64).astype(np.uint32)
    rows[:, 9] &= 0xFFFFF000                                # 150 bases: bits beyond the read are zero in a packed row
    q = rng.integers(2, 64, (n, 150)).astype(np.uint8)
    head = tmp_path / "chains"
    dfin.write_df(head, rows, q, np.zeros(n, dtype=np.int32), read_len=150, adversarial=1)       # a block per value
    assert (tmp_path / "chains.qualp").stat().st_size > n * 450
    e = Engine(0)
    with dfin.DfFiles(head) as f:
        _check_equal(e, f, head, slab_reads=20_000)
    e.close()


def test_bad_files_are_refused(snk, tmp_path):
    from supernova_amd import dfin
    from supernova_amd.engine import Engine
    from supernova_amd.lib import SnkError
    rows, lens, q, bc = _random_triple(3000, 3)
    lens[:] = 150
    q[q == 0] = 9
    head = tmp_path / "r"
    dfin.write_df(head, rows, q, bc, lens=lens, read_len=150)
    e = Engine(0)
    fb, qp = (tmp_path / "r.fastb").read_bytes(), (tmp_path / "r.qualp").read_bytes()
    import struct
    var = struct.unpack_from("<Q", qp, 8)[0]
    # a quality chain cut short: the terminator of read 1000 and its last block's tail overwritten by a long block header
    o = struct.unpack_from("<Q", qp, var + 8 * 1001)[0]
    bad = bytearray(qp)
    bad[o - 1] = 200
    (tmp_path / "r.qualp").write_bytes(bytes(bad))
    with dfin.DfFiles(head) as f, pytest.raises(SnkError, match="truncated quality block|more quality values"):
        f.ingest(e)
    (tmp_path / "r.qualp").write_bytes(qp)
    # a length that its bytes cannot hold
    fvar, ffix = struct.unpack_from("<QQ", fb, 8)
    bad = bytearray(fb)
    struct.pack_into("<I", bad, ffix + 4 * 77, 9999)
    (tmp_path / "r.fastb").write_bytes(bytes(bad))
    with dfin.DfFiles(head) as f, pytest.raises(SnkError, match="bad length"):
        f.ingest(e, read_len=150)
    # an offset table that runs backwards
    bad = bytearray(fb)
    struct.pack_into("<Q", bad, fvar + 8 * 500, 24)
    (tmp_path / "r.fastb").write_bytes(bytes(bad))
    with dfin.DfFiles(head) as f, pytest.raises(SnkError, match="offset"):
        f.ingest(e, read_len=150, slab_reads=64)
    (tmp_path / "r.fastb").write_bytes(fb)
    # mismatched read counts, a file that is not feudal, an index of another read set
    dfin.write_df(tmp_path / "s", rows[:10], q[:10], bc[:10], read_len=150)
    with pytest.raises(SnkError, match="holds 10 reads"):
        (tmp_path / "x.fastb").write_bytes(fb); (tmp_path / "x.qualp").write_bytes((tmp_path / "s.qualp").read_bytes()); (tmp_path / "x.bci").write_bytes((tmp_path / "r.bci").read_bytes())
        dfin.DfFiles(tmp_path / "x")
    with pytest.raises(SnkError, match="indexes 10 reads"):
        (tmp_path / "x.qualp").write_bytes(qp); (tmp_path / "x.bci").write_bytes((tmp_path / "s.bci").read_bytes())
        dfin.DfFiles(tmp_path / "x")
    with pytest.raises(SnkError):
        (tmp_path / "y.fastb").write_bytes(b"\0" * 10); (tmp_path / "y.qualp").write_bytes(qp)
        dfin.DfFiles(tmp_path / "y", with_bci=False)
    # the good files still decode after all that
    with dfin.DfFiles(head) as f:
        _check_equal(e, f, head)
    e.close()


def test_streamed_count_graph_equals_resident(snk, tmp_path):
    """snk_dev_ingest_df_count_graph: slabs -> streamed job == one resident call on the reads the same files decode to (table, counts,
    contexts, spectrum, unitigs), whole file and as two 'ranks' halves each against its own resident call; ign_bc_below honoured"""
    from supernova_amd import dfin, synth
    from supernova_amd.engine import Engine, Params
    n = 120_000
    sp = synth.synth_params(n, seed=0x5EED0DF1, unbarcoded_ppm=0)
    head = tmp_path / "reads"
    dfin.write_synth_df(head, sp, qual_jitter=8)
    e = Engine(0)
    with dfin.DfFiles(head) as f:
        for first, cnt, ign in ((0, n, 0), (0, n // 2, 0), (n // 2, n // 2, 0), (0, n, 50_000)):
            dr = f.ingest(e, first=first, n=cnt)
            reads = dr.dev_reads()
            reads.ign_bc_below = ign
            reads.read_index_base = first
            ref = e.count_graph_reads(reads, Params(K=48))
            want = (ref.unitigs(), ref.keys(), ref.counts(), ref.ctx(), ref.spectrum(), ref.good_len())
            dr.close()
            for slab, mode in ((0, 2), (7000, 2), (0, 0), (7000, 0), (5000, 1)):
                e.set_option("df_stream", mode)                 # 2 streamed, 0 compact, 1 the library's choice
                res, st = f.count_graph(e, Params(K=48), first=first, n=cnt, slab_reads=slab, ign_bc_below=ign)
                assert st["mode"] == ("streamed" if mode == 2 else "compact")
                assert st["n_reads"] == cnt and res.n_reads == cnt and st["n_slabs"] == -(-cnt // (slab or 262144))
                assert res.unitigs() == want[0] and np.array_equal(res.keys(), want[1]) and np.array_equal(res.counts(), want[2])
                assert np.array_equal(res.ctx(), want[3]) and np.array_equal(res.spectrum(), want[4]) and np.array_equal(res.good_len(), want[5])
        e.clear_option("df_stream")
    e.close()
    # the default is the compact form (the resident step may partition twice and picks its count kernel); the streamed job on request
    e = Engine(0)
    with dfin.DfFiles(head) as f:
        r1, s1 = f.count_graph(e, Params(K=48))
        u1 = r1.unitigs()
        e.set_option("df_stream", 2)
        r2, s2 = f.count_graph(e, Params(K=48))
        assert (s1["mode"], s2["mode"]) == ("compact", "streamed") and r2.unitigs() == u1
    e.close()


def test_per_barcode_graphs_from_the_stage_inputs(snk, tmp_path):
    """SNK_F_GROUPED through the DF seam: the group of a read is its barcode's ordinal in reads.bci -- the same (group, k-mer) table, counts
    and unitigs with their groups as the resident grouped call on the arrays the files decode to (BASELINE config 5's shape); without the
    barcode index the call is refused."""
    from supernova_amd import dfin, synth
    from supernova_amd.engine import Engine, Params
    from supernova_amd.lib import SnkError
    n = 160_000
    sp = synth.synth_params(n, seed=0x5EED0DF5, unbarcoded_ppm=0)
    head = tmp_path / "reads"
    dfin.write_synth_df(head, sp, qual_jitter=8)
    e = Engine(0)
    prm = Params(K=48, grouped=True, min_bc=0, min_freq=2, sorted_table=False)
    with dfin.DfFiles(head) as f:
        dr = f.ingest(e)
        reads = dr.dev_reads()
        reads.group, reads.bc = reads.bc, None          # the barcode ordinals ARE the groups
        def sig(r):          # (the table is left in bucket order, and the bucket count is the call's own choice: compared as a set)
            k, c = r.keys(), r.counts()
            o = np.lexsort(tuple(k[:, j] for j in range(k.shape[1] - 1, -1, -1)))
            ug = sorted(zip(r.unitigs(), r.unitig_groups().tolist()))
            return r.n_kmers, k[o].tobytes(), c[o].tobytes(), ug
        ref = e.count_graph_reads(reads, prm)
        want = sig(ref)
        assert want[0] > 1000
        del ref
        dr.close()
        res, st = f.count_graph(e, prm)
        assert st["mode"] == "compact"
        assert sig(res) == want
    with dfin.DfFiles(head, with_bci=False) as f, pytest.raises(SnkError, match="barcode index"):
        f.count_graph(e, prm)
    e.close()


@pytest.mark.skipif(not refio.REF_DRIVER.exists(), reason="oracle/_ref/snref_driver not built")
def test_reference_written_triple_1m(snk, tmp_path):
    """a 1.05 M-read triple written by the REFERENCE's writers on this box (vecbvec::WriteAll, VecPQVec store, BinaryWriter; snref_driver ...
    formats): ragged lengths, noisy qualities (the reference's optimal block chains: every width, multi-block), device == host readers"""
    from supernova_amd import dfin, synth
    from supernova_amd.engine import Engine
    n = 1_050_000
    rows, lens, q, bc = _random_triple(n, 2024, clip=63)      # (the reference's encoder refuses values above 63, PQVec.cc:30-35)
    bases = synth.unpack_rows(rows, 150)
    refio.write_snkrd(tmp_path / "in.snkrd", lens, synth.codes_to_ascii(bases), q, bc)
    out = refio.run_ref(tmp_path / "in.snkrd", tmp_path / "out", threads=32, mode="formats")
    assert f"reads={n}" in out
    head = tmp_path / "out" / "reads"
    e = Engine(0)
    with dfin.DfFiles(head) as f:
        assert f.n_reads == n
        _check_equal(e, f, head)
        _check_equal(e, f, head, first=n // 3 + 1, n=n // 3, slab_reads=50_000, threads=7)
    # and against what went in
    hr, hq, hl, hb, mx = _host_arrays(head, n, 160)
    assert np.array_equal(hr, rows) and np.array_equal(hl, lens) and np.array_equal(hq[:, :150], q) and np.array_equal(hb, bc)
    e.close()


def test_tuning_struct_and_options(snk, monkeypatch):
    """snk_ctx_set_tuning / snk_ctx_get_tuning / snk_ctx_set_option (include/snk.h): the count kernel pinned through the struct is the one
    that runs and is echoed back; unknown names and out-of-range fields are refused; SNK_TUNING is applied when a context is created."""
    from supernova_amd.engine import Engine, Params
    from supernova_amd.lib import SnkError
    c = goldens.load("synth_2k_err")
    import torch
    dev = torch.device("cuda", 0)
    rows = torch.from_numpy(c.rows.view(np.int32)).to(dev); quals = torch.from_numpy(np.ascontiguousarray(c.quals)).to(dev)
    bc = torch.from_numpy(c.bc.astype(np.int32)).to(dev); lens = torch.from_numpy(c.lens.astype(np.uint16).view(np.int16)).to(dev)
    e = Engine(0)
    run = lambda: e.count_graph(rows, c.read_len, quals=quals, bc=bc, lens=lens, params=Params(K=48), ign_bc_below=c.ign_bc_below)
    want = run().unitigs()
    assert e.get_tuning()["last_count_kernel"] == 1 and e.get_tuning()["last_count_limit"] == 1216 and e.get_tuning()["count_kernel"] == 0
    for kernel, slots, limit in ((2, 0, 1920), (2, 1000, 1000), (3, 0, 960), (1, 0, 1216)):
        e.set_tuning(count_kernel=kernel, count_tight_slots=slots, target_inst=900)
        assert run().unitigs() == want
        t = e.get_tuning()
        assert (t["count_kernel"], t["last_count_kernel"], t["last_count_limit"], t["target_inst"]) == (kernel, kernel, limit, 900), t
    e.set_tuning()                                  # everything back to the library's choice
    assert e.get_option("count_tight") is None and e.get_option("target_inst") is None and e.get_tuning()["count_kernel"] == 0
    e.set_option("minimiser_len", 20)
    assert run().unitigs() == want and e.get_tuning()["last_minimiser_len"] == 20 and e.get_option("minimiser_len") == 20
    e.clear_option("minimiser_len")
    assert run().unitigs() == want and e.get_tuning()["last_minimiser_len"] == 16
    with pytest.raises(SnkError, match="no option"):
        e.set_option("no_such_knob", 1)
    with pytest.raises(SnkError):
        e.set_tuning(count_kernel=9)
    with pytest.raises(KeyError):
        e.get_option("no_such_knob")
    assert "count_tight" in e.options() and len(e.options()) >= 40
    e.close()
    monkeypatch.setenv("SNK_TUNING", "count_tight=1500,hot_min=77")
    e = Engine(0)
    assert e.get_option("count_tight") == 1500 and e.get_option("hot_min") == 77 and e.get_tuning()["count_tight_slots"] == 1500
    e.close()
    monkeypatch.setenv("SNK_TUNING", "count_tight=1500,bogus=1")
    with pytest.raises(SnkError, match="SNK_TUNING"):
        Engine(0)


@pytest.mark.skipif(not refio.REF_DRIVER.exists(), reason="oracle/_ref/snref_driver not built")
def test_df_seam_end_to_end_vs_reference_200k(snk, tmp_path):
    """The whole seam against the reference itself, nothing of this repo in between: the reference's writers make reads.fastb / .qualp / .bci
    (snref_driver ... formats), the reference's createDict + buildEdges make the table and the unitigs of the same reads (snref_driver ...
    dump); the device decodes the files slab by slab into a streamed job -- good lengths, retained table, contexts, spectrum, unitigs and the
    .bv file of snk_mspedges equal the reference's, bit for bit.  Reads sorted by barcode (what a .bci index needs), ragged lengths, Q2 tails."""
    import os
    import subprocess
    from supernova_amd import dfin, graphio, synth
    from supernova_amd.engine import Engine, Params
    n = 200_000
    sp = synth.synth_params(n, seed=0x5EED0D5E)
    rows, quals, bc = synth.synth_host(sp)
    order = np.argsort(bc, kind="stable")                       # the model's unbarcoded pairs are scattered: the index wants them first
    order = order.reshape(-1)                                   # (stable: mates stay next to each other, pairs keep their order)
    rows, quals, bc = rows[order], quals[order], bc[order]
    rng = np.random.default_rng(7)
    lens = np.full(n, 150, dtype=np.uint16)
    short = rng.random(n) < 0.05
    lens[short] = rng.integers(40, 150, int(short.sum()))
    codes = synth.unpack_rows(rows, 150)
    codes[np.arange(150)[None, :] >= lens[:, None]] = 0
    quals = quals.copy()
    quals[np.arange(150)[None, :] >= lens[:, None]] = 0
    refio.write_snkrd(tmp_path / "in.snkrd", lens, synth.codes_to_ascii(codes), quals, bc)
    th = min(32, os.cpu_count() or 8)
    refio.run_ref(tmp_path / "in.snkrd", tmp_path / "fmt", threads=th, mode="formats")
    refio.run_ref(tmp_path / "in.snkrd", tmp_path / "out", threads=th)
    d = refio.read_ref_dump(tmp_path / "out")
    hist = np.asarray(d["hist"]["vals"], dtype=np.int64)
    e = Engine(0)
    with dfin.DfFiles(tmp_path / "fmt" / "reads") as f:
        for slab in (0, 30_000):
            res, st = f.count_graph(e, Params(K=48), slab_reads=slab)
            assert np.array_equal(res.good_len().astype(np.uint32), d["goodlens"])
            k = res.keys()
            assert np.array_equal(k[:, :3], d["kmers"]["k"]) and np.array_equal(np.minimum(res.counts(), (1 << 24) - 1), d["kmers"]["count"])
            assert np.array_equal(res.ctx(), d["kmers"]["ctx"]) and res.unitigs() == d["unitigs"]
            spec = res.spectrum()
            nz = np.nonzero(spec)[0]
            assert np.array_equal(spec[: nz[-1] + 1].astype(np.int64), hist)
    e.close()
    exe = Path(__file__).resolve().parent.parent / "supernova_amd" / "bin" / "snk_mspedges"
    out = tmp_path / "asm_graph.bv"
    r = subprocess.run([str(exe), f"LR={tmp_path / 'fmt' / 'reads.fastb'}", f"OUT={out}", "SLAB_READS=50000", "IO_THREADS=3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    off, bases = graphio.read_bv(str(out))
    assert sorted(graphio.arrays_to_unitigs(off, bases)) == sorted(d["unitigs"])


def test_corrupted_files_never_get_past_the_decoder(snk, tmp_path):
    """Random damage to the offset tables, the length table, the quality bytes and the control blocks: every call either fails with an error
    of the library (SNK_E_IO / SNK_E_ARG / SNK_E_UNSUPPORTED) or decodes -- and what it decodes is what the host readers decode from the same
    bytes (damage inside a quality block's values is still a valid file).  Nothing hangs, nothing reads outside its buffers."""
    import struct
    from supernova_amd import dfin, formats
    from supernova_amd.engine import Engine
    from supernova_amd.lib import SnkError
    rows, lens, q, bc = _random_triple(4000, 17)
    head = tmp_path / "r"
    dfin.write_df(head, rows, q, bc, lens=lens, read_len=150)
    good = {ext: (tmp_path / f"r.{ext}").read_bytes() for ext in ("fastb", "qualp")}
    rng = np.random.default_rng(99)
    e = Engine(0)
    outcomes = {"error": 0, "decoded": 0}
    for trial in range(40):
        ext = "fastb" if trial % 2 else "qualp"
        raw = bytearray(good[ext])
        var, fix = struct.unpack_from("<QQ", raw, 8)
        kind = trial % 5
        if kind == 0:   # an offset
            at = var + 8 * int(rng.integers(0, 4001))
            struct.pack_into("<Q", raw, at, int(rng.integers(0, 1 << 40)))
        elif kind == 1 and ext == "fastb":   # a length
            struct.pack_into("<I", raw, fix + 4 * int(rng.integers(0, 4000)), int(rng.integers(0, 1 << 20)))
        elif kind == 2:   # the control block
            raw[int(rng.integers(0, 24))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 3:   # truncate
            del raw[int(rng.integers(24, len(raw))):]
        else:             # a data byte
            for _ in range(8):
                raw[int(rng.integers(24, var))] = int(rng.integers(0, 256))
        (tmp_path / f"r.{ext}").write_bytes(bytes(raw))
        try:
            with dfin.DfFiles(head) as f:
                dr = f.ingest(e, read_len=160, slab_reads=int(rng.choice([0, 333, 1000])))
                got = _device_arrays(e, dr)
                dr.close()
            hr, hl, _ = formats.read_fastb(str(head) + ".fastb")
            hq = formats.read_qualp(str(head) + ".qualp", 4000, 160)
            assert np.array_equal(got[0][:, :hr.shape[1]], hr) and np.array_equal(got[2], hl) and np.array_equal(got[1], hq)
            outcomes["decoded"] += 1
        except SnkError as ex:
            assert ex.code in (-1, -5, -6), ex
            outcomes["error"] += 1
        (tmp_path / f"r.{ext}").write_bytes(good[ext])
    assert outcomes["error"] >= 10 and outcomes["decoded"] >= 3, outcomes
    with dfin.DfFiles(head) as f:
        _check_equal(e, f, head)
    e.close()


def test_long_reads_250(snk, tmp_path):
    """rows of 16 words (reads up to 256 bases: the quality rows leave the fused trim's 160-byte limit, the streamed job takes the row-wise
    trim): decode == host readers on ragged 250-base triples, and the streamed count+graph == a resident call on 250-base synthetic reads"""
    from supernova_amd import dfin, synth
    from supernova_amd.engine import Engine, Params
    n = 20_001
    rows, lens, q, bc = _random_triple(n, 250, max_len=250)
    head = tmp_path / "long"
    dfin.write_df(head, rows, q, bc, lens=lens, read_len=250, adversarial=0x250)
    e = Engine(0)
    with dfin.DfFiles(head) as f:
        assert f.max_len(e) == 250
        _check_equal(e, f, head, slab_reads=3001)
        _check_equal(e, f, head, read_len=256)
    m = 60_000
    sp = synth.synth_params(m, seed=0x5EED0250, unbarcoded_ppm=0, read_len=250)
    dfin.write_synth_df(tmp_path / "s250", sp, qual_jitter=4)
    with dfin.DfFiles(tmp_path / "s250") as f:
        dr = f.ingest(e)
        assert dr.read_len == 250 and int(dr.raw.row_words) == 16
        ref = e.count_graph_reads(dr.dev_reads(), Params(K=48))
        want = (ref.unitigs(), ref.keys(), ref.counts(), ref.ctx())
        assert len(want[0]) > 0
        dr.close()
        res, st = f.count_graph(e, Params(K=48), slab_reads=9000)
        assert res.unitigs() == want[0] and np.array_equal(res.keys(), want[1]) and np.array_equal(res.counts(), want[2]) and np.array_equal(res.ctx(), want[3])
    e.close()


def test_trimmed_compact_form(snk, tmp_path):
    """snk_dev_ingest_df_trimmed: rows and barcode ids as the full decode's, good lengths == the trim of the full decode's quality rows (and the
    oracle's GoodLenTailFinder restatement), no quality rows on the device; count+graph on the compact form == on the full form; rank slices."""
    import oracle_lib
    from supernova_amd import dfin
    from supernova_amd.engine import Engine, Params
    rows, lens, q, bc = _random_triple(30_011, 41)
    q[:, :] = np.where(np.random.default_rng(3).random(q.shape) < 0.04, 3, np.maximum(q, 8))        # a few low qualities, everything else above min_qual
    q[np.arange(150)[None, :] >= lens[:, None]] = 0
    head = tmp_path / "t"
    dfin.write_df(head, rows, q, bc, lens=lens, read_len=150)
    e = Engine(0)
    with dfin.DfFiles(head) as f:
        n = f.n_reads
        for K in (48, 60):
            want_gl = oracle_lib.good_lens(q, lens, K=K) if "K" in oracle_lib.good_lens.__code__.co_varnames else None
            for first, cnt, slab in ((0, n, 0), (10_001, 9_999, 1234)):
                dr = f.ingest_trimmed(e, K=K, min_qual=7, first=first, n=cnt, slab_reads=slab)
                assert not dr.raw.quals and not dr.raw.lens and dr.raw.good_len
                got_rows = _dl(e, dr.raw.rows, (cnt, int(dr.raw.row_words)), np.uint32)
                got_gl = _dl(e, dr.raw.good_len, (cnt,), np.uint16)
                got_bc = _dl(e, dr.raw.bc, (cnt,), np.int32)
                assert np.array_equal(got_rows, rows[first:first + cnt]) and np.array_equal(got_bc, bc[first:first + cnt])
                full = f.ingest(e, first=first, n=cnt)
                r_full = e.count_graph_reads(full.dev_reads(), Params(K=K))
                assert np.array_equal(got_gl, r_full.good_len())
                if want_gl is not None:
                    assert np.array_equal(got_gl.astype(np.uint32), want_gl[first:first + cnt])
                u_full, k_full = r_full.unitigs(), r_full.keys()
                full.close()
                r_c = e.count_graph_reads(dr.dev_reads(), Params(K=K))
                assert r_c.unitigs() == u_full and np.array_equal(r_c.keys(), k_full)
                dr.close()
    e.close()
