"""Host-side product functions behind the C ABI that need no GPU: the .bv hand-off file and the graph-from-unitigs
step (buildHBVFromEdges), checked against the reference's golden vectors."""
import numpy as np
import pytest

import goldens


@pytest.mark.parametrize("name", goldens.CASES)
def test_hbv_from_unitigs_matches_reference(snk, name):
    from supernova_amd import graphio
    c = goldens.load(name)
    off, bases = graphio.unitigs_to_arrays(c.exp_unitigs)
    h = graphio.hbv_from_unitigs(48, off, bases)
    assert graphio.hbv_text(c.exp_unitigs, h) == c.exp_hbv


@pytest.mark.parametrize("name", goldens.K60_CASES)
def test_hbv_k60_matches_reference(snk, name):
    from supernova_amd import graphio
    g = goldens.Case60(name)
    off, bases = graphio.unitigs_to_arrays(g.exp_unitigs)
    assert graphio.hbv_text(g.exp_unitigs, graphio.hbv_from_unitigs(60, off, bases)) == g.exp_hbv


def test_bv_roundtrip_and_layout(snk, tmp_path):
    """.bv: "BINWRITE", u64 count, per entry u32 length + ceil(len/4) bytes, base j at bits 2*(j%4)
    (lib/tada/src/debruijn.rs:895-929; round trip as in sim_tests.rs:142-179)."""
    from supernova_amd import graphio
    c = goldens.load("adversarial")
    off, bases = graphio.unitigs_to_arrays(c.exp_unitigs)
    p = tmp_path / "asm_graph.bv"
    graphio.write_bv(p, off, bases)
    raw = p.read_bytes()
    assert raw[:8] == b"BINWRITE" and int.from_bytes(raw[8:16], "little") == len(c.exp_unitigs)
    l0 = int.from_bytes(raw[16:20], "little")
    assert l0 == len(c.exp_unitigs[0])
    first4 = ["ACGT".index(ch) for ch in c.exp_unitigs[0][:4]]
    assert raw[20] == first4[0] | first4[1] << 2 | first4[2] << 4 | first4[3] << 6
    assert len(raw) == 16 + sum(4 + (len(u) + 3) // 4 for u in c.exp_unitigs)
    off2, bases2 = graphio.read_bv(p)
    assert np.array_equal(off2, off) and np.array_equal(bases2, bases)
    # the oracle's independent writer produces the same bytes
    import ctypes as C
    import oracle_lib
    lib = oracle_lib.load()
    u = oracle_lib.Unitigs()
    u.n = len(c.exp_unitigs)
    u.off = off.ctypes.data_as(C.POINTER(C.c_uint64))
    u.bases = bases.ctypes.data_as(C.POINTER(C.c_uint8))
    p2 = tmp_path / "oracle.bv"
    assert lib.sno_write_bv(str(p2).encode(), C.byref(u)) == 0
    assert p2.read_bytes() == raw


def test_read_bv_rejects_corrupt_headers(snk, tmp_path):
    """The unitig count and every length come from the file: a header that claims more than the bytes hold is an I/O error
    through the C ABI, never an exception or an abort (exit-code contract of the stage: 1, not SIGABRT)."""
    import struct
    from supernova_amd import graphio
    from supernova_amd.lib import SnkError
    good = tmp_path / "g.bv"
    graphio.write_bv(str(good), np.array([0, 5, 9], dtype=np.uint64), np.array([0, 1, 2, 3, 0, 3, 2, 1, 0], dtype=np.uint8))
    raw = good.read_bytes()
    cases = {
        "count": raw[:8] + struct.pack("<Q", 1 << 60) + raw[16:],            # absurd unitig count
        "length": raw[:16] + struct.pack("<I", 0xFFFFFFF0) + raw[20:],       # first unitig claims 4 G bases
        "cut": raw[:-1],                                                     # last payload byte missing
        "magic": b"BINWRITX" + raw[8:],
        "short": raw[:11],
    }
    for name, blob in cases.items():
        f = tmp_path / f"{name}.bv"
        f.write_bytes(blob)
        with pytest.raises(SnkError):
            graphio.read_bv(str(f))
    off, bases = graphio.read_bv(str(good))
    assert list(off) == [0, 5, 9] and list(bases) == [0, 1, 2, 3, 0, 3, 2, 1, 0]


@pytest.mark.parametrize("name,K", [(n, 48) for n in goldens.CASES] + [(n, 60) for n in goldens.K60_CASES])
def test_hbv_files_match_the_reference_writers(snk, tmp_path, name, K):
    """f2: a.hbv (BinaryWriter::writeFile(HyperBasevector)) and a.inv (hbv.Involution, RunStages.cc:418) byte for byte
    against the files the reference binary wrote for the golden cases."""
    from supernova_amd import graphio
    c = goldens.load(name) if K == 48 else goldens.Case60(name)
    off, bases = graphio.unitigs_to_arrays(c.exp_unitigs)
    inv = graphio.write_hbv(tmp_path / "a.hbv", tmp_path / "a.inv", K, off, bases)
    assert (tmp_path / "a.hbv").read_bytes() == c.exp_ahbv
    assert (tmp_path / "a.inv").read_bytes() == c.exp_ainv
    assert np.array_equal(inv[inv], np.arange(len(inv)))          # an involution
