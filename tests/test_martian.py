"""Martian stage adapter: protocol files, FASTH ingest and the barcode indexer KAT (CPU); end-to-end stage run (GPU)."""
import gzip
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import goldens

ROOT = Path(__file__).resolve().parent.parent


def test_bc_indexer_kat():
    """lib/tada/src/utils.rs:435-448 test_bc_indexed."""
    from supernova_amd.martian import BcIndexer
    ix = BcIndexer(["ACGTA\n", "ACGTC\n", "ACGTG\n", "ACGTT\n"])
    assert ix.get_bc_id("ACGTA-1") == 1 and ix.get_bc_id("ACGTA") == 1 and ix.get_bc_id("ACGTA-2") == 5
    assert ix.get_bc_id("ACGTT-1") == 4 and ix.get_bc_id("ACGTT-2") == 8
    assert ix.get_bc_id("AACG-1") is None and ix.get_bc_id("AACG") is None


def _bc_vectors():
    return json.loads((ROOT / "tests" / "golden" / "bc_indexer_vectors.json").read_text())


def test_bc_indexer_literal_vectors():
    """Parity unpinned (Rust): the restated BcIndexer against literal vectors derived by hand from utils.rs:101-164 -- the reference's
    KAT, gem groups up to 255, a duplicated / empty / blank-holding whitelist line, the integer syntax u8::from_str takes, and
    every input the reference panics on."""
    from supernova_amd.martian import BcIndexer
    v = _bc_vectors()
    kat = BcIndexer([l + "\n" for l in v["kat_whitelist"]])
    for q, exp in v["kat"]:
        assert kat.get_bc_id(q) == exp, q
    ix = BcIndexer([l + "\n" for l in v["whitelist"][:-1]] + [v["whitelist"][-1]])        # (the last line has no newline)
    assert ix.num_bcs == len(v["whitelist"])
    for q, exp in v["cases"]:
        if exp == "panic":
            with pytest.raises((ValueError, OverflowError)):
                ix.get_bc_id(q)
        else:
            assert ix.get_bc_id(q) == exp, q
    for f, exp in v["fastq_fields"]:
        assert ix.get_bc_id(f.split(",")[0] if "," in f else f) == exp, f


@pytest.mark.gpu
def test_device_bc_indexer_literal_vectors():
    """The same vectors through snk_dev_bc_ids (whitelist in HBM, one thread per field): ids equal, every reference panic is an error."""
    from supernova_amd.lib import SnkError
    from supernova_amd.martian import DeviceBcIndexer
    v = _bc_vectors()

    def fields_of(strs, F=64):
        a = np.zeros((len(strs), F), dtype=np.uint8)
        for i, s in enumerate(strs):
            b = s.encode()[:F]
            a[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        return a

    dev = DeviceBcIndexer(("\n".join(v["whitelist"])).encode())
    assert dev.num_bcs == len(v["whitelist"])
    good = [(q, e) for q, e in v["cases"] + v["fastq_fields"] if e != "panic"]
    got = dev.ids_of_fields(fields_of([q for q, _ in good])).tolist()
    assert got == [e or 0 for _, e in good], [(q, e, g) for (q, e), g in zip(good, got) if (e or 0) != g]
    for q, e in v["cases"]:
        if e == "panic":
            with pytest.raises(SnkError):
                dev.ids_of_fields(fields_of([q]))
    dev.close()
    kat = DeviceBcIndexer(("\n".join(v["kat_whitelist"])).encode())
    assert kat.ids_of_fields(fields_of([q for q, _ in v["kat"]])).tolist() == [e or 0 for _, e in v["kat"]]
    kat.close()


@pytest.mark.gpu
def test_device_bc_indexer_matches_host_and_kat():
    """f3: barcode ids on the device == BcIndexer (utils.rs:101-164): the reference's KAT (:435-448), duplicate whitelist
    lines (last wins), CRLF, gem groups, raw-barcode suffixes, non-ACGT and over-long sequences, 200 k random fields; the
    reference's two panics come back as errors."""
    from supernova_amd.martian import BcIndexer, DeviceBcIndexer
    from supernova_amd.lib import SnkError

    def fields_of(strs, F=64):
        a = np.zeros((len(strs), F), dtype=np.uint8)
        for i, s in enumerate(strs):
            b = s.encode()[:F]
            a[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        return a

    kat = DeviceBcIndexer(b"ACGTA\nACGTC\nACGTG\nACGTT")
    q = ["ACGTA-1", "ACGTA", "ACGTA-2", "ACGTT-1", "ACGTT-2", "AACG-1", "AACG", "ACGTA-1,NNNNN", "ACGTA-2-x"]
    assert kat.ids_of_fields(fields_of(q)).tolist() == [1, 1, 5, 4, 8, 0, 0, 1, 5]
    for bad in ("ACGTA-", "ACGTA-x", "ACGTA-300", "ACGTA-0"):
        with pytest.raises(SnkError):
            kat.ids_of_fields(fields_of([bad]))
    kat.close()

    rng = np.random.default_rng(11)
    wl = ["".join("ACGT"[i] for i in rng.integers(0, 4, 16)) for _ in range(50_000)]
    wl[777] = wl[3]                         # duplicate line: the later index wins
    wl[900] = "ACGTNNNNACGTNNNN"             # any characters are allowed in a whitelist line
    text = ("\r\n".join(wl[:100]) + "\r\n" + "\n".join(wl[100:]) + "\n").encode()
    host = BcIndexer([l + "\n" for l in wl])
    dev = DeviceBcIndexer(text)
    assert dev.num_bcs == host.num_bcs == len(wl)
    qs = []
    for _ in range(200_000):
        r = rng.random()
        s = wl[int(rng.integers(0, len(wl)))] if r < 0.8 else "".join("ACGTN"[i] for i in rng.integers(0, 5, int(rng.integers(1, 40))))
        if rng.random() < 0.02:
            s = s[:-1] + "N"
        t = rng.random()
        if t < 0.5:
            s += f"-{int(rng.integers(1, 9))}"
        if rng.random() < 0.3:
            s += ",RAWBARCODE"
        qs.append(s)
    qs += [wl[3], wl[777] + "-3", wl[900] + "-2"]
    exp = np.array([host.get_bc_id(s.split(",")[0]) or 0 for s in qs], dtype=np.int32)
    got = dev.ids_of_fields(fields_of(qs))
    assert np.array_equal(got, exp)
    assert got[-3] == 777 + 1 and (exp > 0).sum() > 100_000
    dev.close()


def make_fasth(c, tmp_path, n_files=2):
    """Golden synthetic case -> FASTH files (R1 = read 2q, R2 = read 2q+1 share the barcode) + whitelist."""
    from supernova_amd import synth
    rng = np.random.default_rng(5)
    nbc = int(c.bc.max())
    wl = set()
    while len(wl) < nbc + 3:
        wl.add("".join("ACGT"[i] for i in rng.integers(0, 4, 16)))
    wl = sorted(wl)
    wpath = tmp_path / "whitelist.txt"
    wpath.write_text("\n".join(wl) + "\n")
    asc = synth.codes_to_ascii(c.codes)
    n = c.rows.shape[0]
    files = []
    per = (n // 2 + n_files - 1) // n_files
    for fi in range(n_files):
        p = tmp_path / f"chunk{fi}.fasth.gz"
        with gzip.open(p, "wt") as f:
            for q in range(fi * per, min((fi + 1) * per, n // 2)):
                r1, r2 = 2 * q, 2 * q + 1
                assert c.bc[r1] == c.bc[r2]
                b = int(c.bc[r1])
                bcs = (wl[b - 1] + "-1,RAWRAWRAW") if b > 0 else "NNNNNNNNNNNNNNNN"
                f.write(f"@pair{q}\n")
                for r in (r1, r2):
                    L = int(c.lens[r])
                    f.write(asc[r, :L].tobytes().decode() + "\n")
                    f.write((c.quals[r, :L] + 33).tobytes().decode() + "\n")
                f.write(bcs + "\n" + "F" * 16 + "\nACGTACGT\nFFFFFFFF\n")
        files.append(str(p))
    return files, str(wpath)


def test_fasth_ingest(tmp_path):
    from supernova_amd.martian import BcIndexer, read_fasth
    c = goldens.load("synth_2k_err")
    files, wl = make_fasth(c, tmp_path)
    asc, quals, lens, bc = read_fasth(files, BcIndexer.from_file(wl))
    from supernova_amd import synth
    assert np.array_equal(synth.ascii_to_codes(asc), c.codes)
    assert np.array_equal(quals, c.quals) and np.array_equal(lens, c.lens)
    # ids are whitelist positions + 1: injective relabelling of the golden ids, 0 stays 0
    assert np.array_equal(bc == 0, c.bc == 0)
    m = {}
    for a, b in zip(c.bc, bc):
        assert m.setdefault(int(a), int(b)) == int(b)
    assert len(set(m.values())) == len(m)


def test_native_fasth_reader_matches_python(snk, tmp_path):
    """snk_read_fasth (C++/zlib, CPU-only code of libsnk) == the Python restatement of MultiFastqIter on the same files:
    bases, qualities, lengths and the barcode fields (part before the first ','), gzip and plain text, ragged lengths."""
    from supernova_amd.martian import BcIndexer, read_fasth, read_fasth_native
    c = goldens.load("synth_2k_err")
    files, wl = make_fasth(c, tmp_path, n_files=3)
    plain = tmp_path / "plain.fasth"
    plain.write_bytes(gzip.open(files[0], "rb").read())
    # ragged lengths, CRLF line ends, a barcode field without gem group or raw part, an empty read
    rag = tmp_path / "ragged.fasth.gz"
    rng = np.random.default_rng(2)
    with gzip.open(rag, "wt", newline="") as f:
        for q in range(300):
            eol = "\r\n" if q % 3 == 0 else "\n"
            f.write(f"@rag{q}" + eol)
            for _ in range(2):
                L = int(rng.integers(0, 200)) if q else 0
                f.write("".join("ACGTN"[i] for i in rng.integers(0, 5, L)) + eol)
                f.write("".join(chr(33 + int(v)) for v in rng.integers(0, 42, L)) + eol)
            f.write(("ACGTACGTACGTACGT" if q % 2 else "TTTTACGTACGTACGT-2,RAW") + eol + "FFFF" + eol + "ACGT" + eol + "FFFF" + eol)
    for fl in (files, [str(plain)] + files[1:], [str(rag)], files + [str(rag)]):
        asc, qa, lens, fields = read_fasth_native(fl)
        ix = BcIndexer.from_file(wl)
        asc_p, qa_p, lens_p, bc_p = read_fasth([f if f != str(plain) else files[0] for f in fl], ix)
        assert np.array_equal(lens, lens_p) and asc.shape == asc_p.shape
        assert np.array_equal(asc, asc_p) and np.array_equal(qa, qa_p)
        ids = np.array([ix.get_bc_id(bytes(f[:int(np.argmax(f == 0)) if (f == 0).any() else 64]).decode()) or 0 for f in fields], dtype=np.int32)
        assert np.array_equal(np.repeat(ids, 2), bc_p)
    bad = tmp_path / "trunc.fasth"
    bad.write_text("@h\nACGT\nIIII\nACGT\n")
    from supernova_amd.lib import SnkError
    with pytest.raises(SnkError):
        read_fasth_native([str(bad)])
    whole = open(files[0], "rb").read()
    for cut in (len(whole) - 4, len(whole) // 2):
        cutf = tmp_path / f"ncut{cut}.fasth.gz"
        cutf.write_bytes(whole[:cut])
        with pytest.raises(SnkError, match="truncated"):
            read_fasth_native([str(cutf)])


def test_fasth_stream_multi_file(snk, tmp_path):
    """f3: snk_fasth_open/_next -- many files decoded concurrently by a worker pool, batches in any order -- gives, put back into
    file-major order, exactly what the Python restatement of MultiFastqIter reads from the same files; small batches so that every
    file is cut into several; ragged lengths, CRLF, multi-member gzip; plain text is refused like the reference does
    (multifastq.rs "Not a gz file"); a truncated record is an error."""
    from supernova_amd.lib import SnkError
    from supernova_amd.martian import BcIndexer, read_fasth, read_fasth_stream
    c = goldens.load("synth_2k_err")
    files, wl = make_fasth(c, tmp_path, n_files=7)
    rag = tmp_path / "ragged.fasth.gz"
    rng = np.random.default_rng(12)
    with gzip.open(rag, "wt", newline="") as f:
        for q in range(500):
            eol = "\r\n" if q % 3 == 0 else "\n"
            f.write(f"@rag{q}" + eol)
            for _ in range(2):
                L = int(rng.integers(0, 200)) if q else 0
                f.write("".join("ACGTN"[i] for i in rng.integers(0, 5, L)) + eol)
                f.write("".join(chr(33 + int(v)) for v in rng.integers(0, 42, L)) + eol)
            f.write(("ACGTACGTACGTACGT" if q % 2 else "TTTTACGTACGTACGT-2,RAW") + eol + "FFFF" + eol + "ACGT" + eol + "FFFF" + eol)
    # two gzip members in one file (cat a.gz b.gz): one stream for the reader
    multi = tmp_path / "multi.fasth.gz"
    multi.write_bytes(open(files[0], "rb").read() + open(files[1], "rb").read())
    ix = BcIndexer.from_file(wl)
    for fl, threads, bp in ((files, 0, 37), (files + [str(rag)], 3, 64), ([str(multi)] + files[2:], 2, 0), ([str(rag)], 1, 1)):
        asc, qa, lens, fields, st = read_fasth_stream(fl, threads=threads, batch_pairs=bp)
        asc_p, qa_p, lens_p, bc_p = read_fasth(fl, ix)
        assert np.array_equal(lens, lens_p) and asc.shape == asc_p.shape
        assert np.array_equal(asc, asc_p) and np.array_equal(qa, qa_p)
        ids = np.array([ix.get_bc_id(bytes(f[:int(np.argmax(f == 0)) if (f == 0).any() else 64]).decode()) or 0 for f in fields], dtype=np.int32)
        assert np.array_equal(np.repeat(ids, 2), bc_p)
        assert st["text_bytes"] == sum(len(gzip.open(p, "rb").read()) for p in fl)
        if bp:
            assert st["batches"] >= sum(-(-n // bp) for n in st["file_pairs"])
    plain = tmp_path / "plain.fasth"
    plain.write_bytes(gzip.open(files[0], "rb").read())
    with pytest.raises(SnkError, match="not a gz file"):
        read_fasth_stream([files[1], str(plain)])
    bad = tmp_path / "trunc.fasth.gz"
    with gzip.open(bad, "wt") as f:
        f.write("@h\nACGT\nIIII\nACGT\n")
    with pytest.raises(SnkError, match="truncated"):
        read_fasth_stream([str(bad)])
    # an interrupted copy: the compressed stream stops before its end.  Even when every record decoded so far is whole (the cut
    # falls inside the gzip trailer, or anywhere at all) the reference's MultiGzDecoder + unwrap() panics; so does the reader
    whole = open(files[0], "rb").read()
    for cut in (len(whole) - 4, len(whole) - 9, len(whole) // 2):
        cutf = tmp_path / f"cut{cut}.fasth.gz"
        cutf.write_bytes(whole[:cut])
        with pytest.raises(SnkError, match="truncated"):
            read_fasth_stream([files[1], str(cutf)])


def test_synth_fasth_writer_round_trip(snk, tmp_path):
    """snk_synth_fasth_write (the generator behind bench.py --ingest) -> the stream reader gives back the synthetic model's reads,
    and the barcode fields are snk_synth_bc_seq(id) + "-1" for barcoded pairs."""
    import ctypes as C
    from supernova_amd import lib as _lib, synth
    from supernova_amd.martian import read_fasth_stream
    lib = _lib.load()
    sp = synth.synth_params(6000, seed=77)
    rows, quals, bc = synth.synth_host(sp)
    err = C.create_string_buffer(512)
    paths, tot = [], 0
    for fi, (a, n) in enumerate(((0, 1000), (1000, 1500), (2500, 500))):
        p = tmp_path / f"s{fi}.fasth.gz"
        tb = C.c_uint64(0)
        assert lib.snk_synth_fasth_write(str(p).encode(), C.byref(sp), a, n, 1, C.byref(tb), err, 512) == 0, err.value
        tot += tb.value
        paths.append(str(p))
    asc, qa, lens, fields, st = read_fasth_stream(paths, threads=3, batch_pairs=400)
    assert st["text_bytes"] == tot and asc.shape[0] == 6000
    assert np.array_equal(synth.ascii_to_codes(asc), synth.unpack_rows(rows, sp.read_len))
    assert np.array_equal(qa, quals[:, :sp.read_len]) and np.all(lens == sp.read_len)
    for q in (0, 1, 17, 2999):
        b = int(bc[2 * q])
        f = bytes(fields[q]).rstrip(b"\0").decode()
        if b > 0:
            buf = C.create_string_buffer(16)
            lib.snk_synth_bc_seq(b, buf)
            assert f == buf.raw.decode() + "-1"
        else:
            assert f == "NNNNNNNNNNNNNNNN-1"


def _run_stage(tmp_path, stage_type, args=None, outs=None, extra=None, env=None):
    md = tmp_path / f"md_{stage_type}"
    md.mkdir()
    files = tmp_path / "files"
    files.mkdir(exist_ok=True)
    if args is not None:
        (md / "_args").write_text(json.dumps(args))
    if outs is not None:
        (md / "_outs").write_text(json.dumps(outs))
    for k, v in (extra or {}).items():
        (md / k).write_text(json.dumps(v))
    run_file = str(tmp_path / f"run_{stage_type}")
    e = dict(os.environ, PYTHONPATH=str(ROOT))
    e.update(env or {})
    r = subprocess.run([sys.executable, "-m", "supernova_amd.martian", "martian", "asm_sn_gpu", stage_type, str(md), str(files), run_file],
                       capture_output=True, text=True, env=e, cwd=str(ROOT))
    return r, md, run_file


def test_protocol_split_join_and_errors(tmp_path):
    r, md, run_file = _run_stage(tmp_path, "split", args={"fastqs": [], "barcode_whitelist": "x", "min_kmer_obs": 3})
    assert r.returncode == 0, r.stderr
    sd = json.loads((md / "_stage_defs").read_text())
    assert len(sd["chunks"]) == 1 and "__mem_gb" in sd["chunks"][0]
    assert (md / "_complete").exists() and Path(run_file + ".split_stage_defs").exists() and Path(run_file + ".split_complete").exists()
    r, md, run_file = _run_stage(tmp_path, "join", args={}, outs={"asm_graph": None},
                                 extra={"_chunk_defs": [{}], "_chunk_outs": [{"asm_graph": "/x/asm_graph.bv"}]})
    assert r.returncode == 0, r.stderr
    assert json.loads((md / "_outs").read_text())["asm_graph"] == "/x/asm_graph.bv"
    # main without inputs: failure lands in _errors, no _complete (lib.rs:568-602)
    r, md, run_file = _run_stage(tmp_path, "main", args={"fastqs": ["/nonexistent.gz"], "barcode_whitelist": "/nonexistent"}, outs={})
    assert r.returncode != 0 and (md / "_errors").exists() and not (md / "_complete").exists()


@pytest.mark.gpu
def test_stage_main_end_to_end(tmp_path):
    """FASTH + whitelist -> stage main -> asm_graph.bv == the reference's unitigs for the same reads."""
    from supernova_amd import graphio
    c = goldens.load("synth_2k_err")
    files, wl = make_fasth(c, tmp_path)
    outp = str(tmp_path / "files" / "asm_graph.bv")
    r, md, run_file = _run_stage(tmp_path, "main", args={"fastqs": files, "barcode_whitelist": wl, "min_kmer_obs": 3, "trim_min_qual": 7},
                                 outs={"asm_graph": outp})
    assert r.returncode == 0, (r.stdout, r.stderr, (md / "_errors").read_text() if (md / "_errors").exists() else "")
    assert (md / "_complete").exists()
    off, bases = graphio.read_bv(json.loads((md / "_outs").read_text())["asm_graph"])
    assert graphio.arrays_to_unitigs(off, bases) == c.exp_unitigs


@pytest.mark.gpu
def test_host_api_count_graph_and_hbv():
    """snk_count_graph (host pointers) -> BVComp-ordered unitigs -> snk_hbv_from_unitigs == the reference's HBV."""
    from supernova_amd import graphio, synth
    from supernova_amd.martian import count_graph_host
    c = goldens.load("adversarial")
    asc = synth.codes_to_ascii(c.codes)
    bc = c.bc.copy()
    # the host API has no ign_bc_below plumbing in this helper: emulate it with bc = -1 on the first reads
    bc[: c.ign_bc_below] = -1
    off, bases, stats = count_graph_host(asc, c.quals, c.lens, bc)
    us = graphio.arrays_to_unitigs(off, bases)
    assert us == c.exp_unitigs                    # same order as the reference's BVComp sort
    assert stats["n_kmers"] == len(c.exp_keys)
    assert graphio.hbv_text(us, graphio.hbv_from_unitigs(48, off, bases)) == c.exp_hbv
