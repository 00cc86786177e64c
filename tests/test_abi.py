"""CPU-side checks of the C-ABI shared library: it builds for gfx950, loads, exports every symbol that
include/snk.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def test_exports_match_header(snk):
    hdr = (ROOT / "include" / "snk.h").read_text()
    declared = set(re.findall(r"\b(snk_[a-z0-9_]+)\s*\(", hdr))
    from supernova_amd import lib
    bound = set(lib.exported_symbols())
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(snk, name), f"{name} declared in include/snk.h but not exported by libsnk.so"
    assert declared == bound, (declared ^ bound)


def test_no_cpu_fallback(snk):
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    rc = snk.snk_ctx_create(0, C.byref(h), err, 256)
    assert rc == -3 and b"no CPU fallback" in err.value
    import pytest
    from supernova_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(0)


def test_product_never_imports_oracle():
    for f in (ROOT / "supernova_amd").rglob("*"):
        if f.suffix in (".py", ".h", ".hip", ".cpp") and f.is_file():
            txt = f.read_text(errors="ignore")
            assert "snk_oracle" not in txt and "oracle_lib" not in txt and "libsnkoracle" not in txt, f


def test_synth_host_is_deterministic_and_shaped(snk):
    from supernova_amd import synth
    sp = synth.synth_params(4000, seed=123)
    a = synth.synth_host(sp)
    b = synth.synth_host(sp, first=1000, n=500)
    assert np.array_equal(a[0][1000:1500], b[0]) and np.array_equal(a[1][1000:1500], b[1]) and np.array_equal(a[2][1000:1500], b[2])
    rows, quals, bc = a
    assert rows.shape == (4000, 10) and quals.shape == (4000, 150)
    assert 0.005 < (bc == 0).mean() < 0.05
    assert 0.001 < (quals == 12).mean() < 0.004
    # mates of a pair share the barcode
    assert np.array_equal(bc[0::2], bc[1::2])
    # pack/unpack round trip
    codes = synth.unpack_rows(rows, 150)
    assert np.array_equal(synth.pack_rows(codes), rows)


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset) re-runs itself under torch.distributed.run on this node --
    the driver's own N > 1 command line -- and leaves with the ranks' exit code (bench.py::self_launch).  No GPU: the launch is intercepted."""
    import importlib.util
    import sys
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("bench_under_test", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1", "--transport", "gloo"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-8:] == ["--gpus", "4", "--steps", "2", "--warmup", "1", "--transport", "gloo"]
    assert Path(cmd[cmd.index("--master-port") + 2]).name == "bench.py"
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_options_registry_and_no_environment_switches(snk):
    """include/snk.h "tuning": every option has a name and a description; the product reads the environment for tracing and library paths
    only (67 distinct SNK_* switches were read with getenv inside csrc/) -- every algorithmic choice is a context option."""
    names = []
    i = 0
    while snk.snk_option_name(i):
        names.append(snk.snk_option_name(i).decode())
        assert len(snk.snk_option_doc(i)) > 8
        i += 1
    assert len(names) == len(set(names)) >= 40 and {"count_tight", "target_inst", "minimiser_len", "partition_passes", "path_index"} <= set(names)
    assert snk.snk_option_name(i) is None and snk.snk_option_doc(10_000) is None
    env = set()
    for f in (ROOT / "supernova_amd" / "csrc").rglob("*"):
        if f.suffix in (".hip", ".h", ".cc") and f.is_file():
            env |= set(re.findall(r'getenv\("(SNK_[A-Z0-9_]+)"\)', f.read_text(errors="ignore")))
    allowed = {"SNK_TUNING", "SNK_SYNC_TRACE", "SNK_ARENA_TRACE", "SNK_ARENA_POISON", "SNK_INGEST_TRACE", "SNK_HBV_DEPTH", "SNK_RCCL_LIB",
               "SNK_FASTH_LIBDEFLATE", "SNK_FASTH_WHOLE_MAX_MB"}
    assert env <= allowed, env - allowed
    # every option a stage looks up is registered (a look-up of an unknown name aborts at run time: caught here instead)
    looked = set()
    for f in (ROOT / "supernova_amd" / "csrc").glob("*.hip"):
        looked |= set(re.findall(r'snk_opt_(?:u32|u64|is_set)\("([a-z0-9_]+)"', f.read_text()))
        looked |= set(re.findall(r'snk_opt_index\("([a-z0-9_]+)"\)', f.read_text()))
    assert looked <= set(names), looked - set(names)
    t = __import__("supernova_amd.lib", fromlist=["SnkTuning"]).SnkTuning()
    snk.snk_tuning_default(C.byref(t))
    assert t.count_kernel == 0 and t.target_inst == 0


def test_tools_and_bench_compile():
    """bench.py runs tools/r6_full_job.py in a process of its own (config.large_job) and the evidence scripts drive the others: a tool that
    does not even compile must not wait for a GPU box to be noticed."""
    import py_compile
    root = Path(__file__).resolve().parent.parent
    files = [root / "bench.py", root / "__graft_entry__.py"] + sorted((root / "tools").glob("*.py")) + sorted((root / "tests" / "tools").glob("*.py"))
    assert (root / "tools" / "r6_full_job.py") in files
    for f in files:
        py_compile.compile(str(f), doraise=True)
