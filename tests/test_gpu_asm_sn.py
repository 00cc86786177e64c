"""The N-GPU job without a scripting layer: supernova_amd/bin/snk_asm_sn (C++ over the C ABI: snk_shard_step on an RCCL
communicator + snk_shard_gather_unitigs) against the reference's unitigs, and the gather through in-process ranks."""
import hashlib
import json
import subprocess
import threading
from pathlib import Path

import numpy as np
import pytest

import goldens

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "supernova_amd" / "bin" / "snk_asm_sn"


def _run(args, timeout=900):
    r = subprocess.run([str(EXE)] + args, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


def test_asm_sn_fasth_to_bv(snk, tmp_path):
    """FASTH files + whitelist -> asm_graph.bv on a one-rank RCCL communicator == the reference's unitigs in BVComp order."""
    from test_martian import make_fasth
    from supernova_amd import graphio
    c = goldens.load("synth_20k_err")
    files, wl = make_fasth(c, tmp_path, n_files=5)
    out = tmp_path / "asm_graph.bv"
    log = _run([f"FASTH={','.join(files)}", f"WHITELIST={wl}", f"OUT={out}", "WORLD=1", "RANK=0", f"STATS={tmp_path / 'stats.jsonl'}"])
    assert "host read-backs" in log
    off, bases = graphio.read_bv(str(out))
    assert graphio.arrays_to_unitigs(off, bases) == c.exp_unitigs
    st = json.loads((tmp_path / "stats.jsonl").read_text().splitlines()[-1])
    assert st["world"] == 1 and st["reads_rank0"] == c.rows.shape[0]


def test_asm_sn_df_stage_inputs_to_bv(snk, tmp_path):
    """LR=<head>.fastb (+ .qualp, .bci: the ASSEMBLER_DF stage inputs, written by the reference's own writers) -> the rank's byte range is
    decoded on the device (snk_dev_ingest_df) -> one-rank RCCL step -> asm_graph.bv == the reference's unitigs; READ_LEN pins the row length
    every rank of a job would agree on."""
    from supernova_amd import graphio
    fmt = ROOT / "tests" / "golden" / "formats"
    c = goldens.load("synth_2k_err")
    for extra in ([], ["READ_LEN=160"]):
        out = tmp_path / "asm_graph.bv"
        log = _run([f"LR={fmt / 'reads.fastb'}", f"OUT={out}", "WORLD=1", "RANK=0", *extra])
        assert "reads [0, 2000) of 2000" in log and "GB of file bytes" in log
        off, bases = graphio.read_bv(str(out))
        assert graphio.arrays_to_unitigs(off, bases) == c.exp_unitigs


def test_asm_sn_10m_reference_digest(snk, tmp_path):
    """BASELINE config 1 (10 M x 150 bp, seed 0x5EED0001) through the C++ host on a one-rank RCCL communicator: the unitig file's
    content hashes to the reference's own digest (tests/golden/big_hashes.json)."""
    from supernova_amd import graphio
    fix = ROOT / "tests" / "golden" / "big_hashes.json"
    exp = json.loads(fix.read_text())["c1_10m"]
    out = tmp_path / "asm_graph.bv"
    _run([f"SYNTH={exp['n_reads']}", f"SEED={exp['seed']}", f"OUT={out}", "STEPS=2"])
    off, bases = graphio.read_bv(str(out))
    us = graphio.arrays_to_unitigs(off, bases)
    assert len(us) == exp["n_unitigs"] and sum(len(u) for u in us) == exp["unitig_bases"]
    assert us == sorted(us, key=lambda s: (-len(s), s))                # BVComp order in the file
    h = hashlib.sha256()
    for u in sorted(u.encode() for u in us):
        h.update(u)
        h.update(b"\n")
    assert h.hexdigest() == exp["unitigs"]


@pytest.mark.parametrize("W", [1, 3])
def test_gather_unitigs_in_process_ranks(snk, W):
    """snk_shard_gather_unitigs over in-process ranks: root's arrays are the reference's unitigs in BVComp order (the adversarial
    case: circles, a palindrome, unitigs of equal length)."""
    import ctypes as C
    import torch
    from supernova_amd import graphio, lib as _lib
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine, SimWorld
    c = goldens.load("adversarial")
    world = SimWorld(W)
    dev = torch.device("cuda", 0)
    n = c.rows.shape[0]
    bounds = [(n // 2 * r // W) * 2 for r in range(W)] + [n]
    got, errs = {}, []

    def worker(r):
        try:
            e = Engine(0)
            lo, hi = bounds[r], bounds[r + 1]
            sh = ShardedEngine(e, world.comm(r))
            res = sh.count_graph(torch.from_numpy(c.rows[lo:hi].view(np.int32).copy()).to(dev), c.read_len,
                                 quals=torch.from_numpy(np.ascontiguousarray(c.quals[lo:hi])).to(dev),
                                 bc=torch.from_numpy(c.bc[lo:hi].astype(np.int32)).to(dev),
                                 lens=torch.from_numpy(c.lens[lo:hi].astype(np.uint16).view(np.int16)).to(dev),
                                 params=Params(K=48), ign_bc_below=c.ign_bc_below, read_index_base=lo, total_reads=n)
            for root, image in ((0, 0), (W - 1, 32)):
                out = _lib.SnkResult()
                err = C.create_string_buffer(512)
                rc = e.lib.snk_shard_gather_unitigs(e._ctx, sh.comm, C.byref(res.raw), 48, root, image, C.byref(out), e._stream(), err, 512)
                assert rc == 0, err.value
                if r == root:
                    if image:
                        got["image"] = bytes(np.ctypeslib.as_array(out.bv_image, shape=(int(out.bv_bytes),)))
                    else:
                        nu = int(out.n_unitigs)
                        off = np.ctypeslib.as_array(out.unitig_off, shape=(nu + 1,)).copy()
                        bases = np.ctypeslib.as_array(out.unitig_bases, shape=(max(int(off[-1]), 1),))[:int(off[-1])].copy()
                        got["plain"] = graphio.arrays_to_unitigs(off, bases)
                else:
                    assert int(out.n_unitigs) == 0
                e.lib.snk_free(C.byref(out))
            sh.close()
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)
            world.barrier_obj.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errs:
        raise errs[0]
    assert got["plain"] == c.exp_unitigs

    import tempfile
    with tempfile.TemporaryDirectory() as td:
        p = Path(td) / "x.bv"
        p.write_bytes(got["image"])
        off, bases = graphio.read_bv(str(p))
        assert graphio.arrays_to_unitigs(off, bases) == c.exp_unitigs


def test_bench_two_processes_over_gloo_one_gpu(tmp_path):
    """bench.py's N > 1 plumbing and the multi-process step with real message passing between two PROCESSES (two contexts, rank
    slabs, the library's exchange planning, the owner-side join, the self-check against the one-GPU path, max-over-ranks timing,
    one JSON line from rank 0) on a box with one GPU: both ranks share it, the exchanges go through the host over gloo
    (`--transport gloo`).  What it cannot show is RCCL between two devices."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(root / "bench.py"), "--gpus", "2", "--transport", "gloo", "--reads", "3e6", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] and d["value"] > 0
    mg = d["config"]["multi_gpu"]
    assert d["config"]["sharded_self_check"] == "passed" and mg["transport"] == "callbacks" and mg["rccl_ranks"] == 2
    assert mg["exchange_bytes_rank0"]["records"] > 0 and mg["join_ranking"] == "partitioned"


@pytest.mark.parametrize("extra", [[], ["--k", "60"], ["--grouped"]], ids=["c3_k48", "c4_k60", "c5_grouped"])
def test_bench_self_launches_its_ranks(extra):
    """The driver's N = 1 command shape with --gpus 2 and NO launcher around it (WORLD_SIZE unset): bench.py starts its two ranks itself
    (bench.py::self_launch), rank 0 prints the one JSON line, exit code 0.  One line each for the job shapes of BASELINE configs 3, 4 and 5
    (k=48 sharded, k=60 sharded, per-barcode replicas), over `--transport gloo` because this box has one GPU."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--transport", "gloo", "--reads", "2e6", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", *extra]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root), env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] and d["value"] > 0
    assert d["config"]["k"] == (60 if "--k" in extra else 48)
    if "--grouped" in extra:
        assert d["config"]["path"] == "grouped-replicas"
    else:
        assert d["config"]["sharded_self_check"] == "passed" and d["config"]["multi_gpu"]["rccl_ranks"] == 2


@pytest.mark.parametrize("W", [2, 3])
def test_ranks_decode_their_own_byte_ranges_of_the_stage_inputs(snk, W):
    """The N-GPU job at the ASSEMBLER_DF seam, as in-process ranks on one GPU: every rank decodes ITS reads' byte ranges of reads.fastb /
    .qualp / .bci on the device into the compact form (snk_dev_ingest_df_trimmed: rows, good lengths, barcode ids) and runs snk_shard_step on
    it; the gathered unitigs are the reference's (the files are the reference's writers' own, tests/golden/formats)."""
    import ctypes as C
    import torch
    from supernova_amd import dfin, graphio, lib as _lib
    from supernova_amd.engine import Engine, Params
    from supernova_amd.sharded import ShardedEngine, SimWorld
    c = goldens.load("synth_2k_err")
    fmt = ROOT / "tests" / "golden" / "formats" / "reads"
    world = SimWorld(W)
    n = 2000
    bounds = [(n // 2 * r // W) * 2 for r in range(W)] + [n]
    got, errs = {}, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            e = Engine(0)
            lo, hi = bounds[r], bounds[r + 1]
            with dfin.DfFiles(fmt) as f:
                dr = f.ingest_trimmed(e, K=48, min_qual=7, first=lo, n=hi - lo, read_len=150, slab_reads=200)
            reads = dr.dev_reads()
            assert reads.good_len and not reads.quals
            reads.read_index_base = lo
            sh = ShardedEngine(e, world.comm(r))
            res = sh.count_graph_reads(reads, Params(K=48), total_reads=n)
            out = _lib.SnkResult()
            err = C.create_string_buffer(512)
            rc = e.lib.snk_shard_gather_unitigs(e._ctx, sh.comm, C.byref(res.raw), 48, 0, 0, C.byref(out), e._stream(), err, 512)
            assert rc == 0, err.value
            if r == 0:
                nu = int(out.n_unitigs)
                off = np.ctypeslib.as_array(out.unitig_off, shape=(nu + 1,)).copy()
                bases = np.ctypeslib.as_array(out.unitig_bases, shape=(max(int(off[-1]), 1),))[:int(off[-1])].copy()
                got["unitigs"] = graphio.arrays_to_unitigs(off, bases)
            e.lib.snk_free(C.byref(out))
            sh.close()
            dr.close()
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
    assert sorted(got["unitigs"]) == sorted(c.exp_unitigs)
