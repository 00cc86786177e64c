"""ASSEMBLER_DF stage adapter (mro/_assembler_stages.mro:24-39) with the graph built on the MI355X.

Mirror of the reference's stage code mro/stages/denovo/df/__init__.py (Python 2; not importable here):
  split   :8-12     one chunk; the reference reserves 2048 GB / 28 threads for the CPU dictionary -- here 256 GB
                    (deliberate: the stage inputs in host memory; the dictionary lives in HBM).  $SNK_DF_MEM_GB overrides it,
                    e.g. 2048 when the stock DF that follows needs its usual reservation
  main    :81-173   filename-head check (:84-88), DF argv (:123-139), alerts.list + the alarm / Martian::exit (code 185)
                    relay (:139-166, tenkit/supernova/alerts.py), exit-code -> message mapping (:63-79), *.mm files moved
                    into stats/ (:172-173)
  join    :14-15
What changes: when `mspedges` is not supplied by an upstream `_ASM_SN`, this stage computes the unitigs itself
(reads.fastb/.qualp/.bci -> libsnk -> asm_graph.bv) and hands them to the stock `DF` binary through `MSPEDGES=`
(10X/DF.cc:86-207, RunStages.cc:404-413): DF then skips createDict/buildEdges and continues with
buildGraphFromMSP -> read pathing -> ... unchanged.
"""
from __future__ import annotations

import glob
import os
import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np


# Martian hands a `src py` stage record objects (attribute access: args.reads, outs.default, chunk_outs[0].default --
# mro/stages/denovo/df/__init__.py:14-15,81-139); the tests and the C++-side callers use plain dicts.  Both are accepted.
def _get(rec, name, default=None):
    if isinstance(rec, dict):
        return rec.get(name, default)
    return getattr(rec, name, default)


def _set(rec, name, value):
    if isinstance(rec, dict):
        rec[name] = value
    else:
        setattr(rec, name, value)


def split(args):
    return {"chunks": [{"__mem_gb": int(os.environ.get("SNK_DF_MEM_GB", "256")), "__threads": 28, "__special": "asmlarge"}]}


def join(args, outs, chunk_defs, chunk_outs):
    shutil.move(_get(chunk_outs[0], "default"), _get(outs, "default"))


def check_exclude(path: str, ext: str) -> str:
    head, tail = path[:-len(ext)], path[-len(ext):]
    if tail != ext:
        raise Exception("file has incorrect extension: " + path)
    return head


def process_return_code(returncode: int):
    msg = None
    if returncode < 0:
        sig = {-9: "KILL signal", -1: "HUP signal", -2: "INT signal", -15: "TERM signal"}.get(returncode, "signal")
        msg = ("A Supernova process was terminated with a %s (code: %d). This may have been sent by you, your IT admin, "
               "or automatically by the system itself (e.g. the out-of-memory killer)." % (sig, -returncode))
    elif returncode == 99:
        msg = ("Supernova terminated because of insufficient memory. The stage _stdout file may contain additional "
               "useful information, e.g., whether any competing processes were running.")
    return msg


# ---- alerts: the C++ side of DF reports through files, the stage code turns them into Martian calls
# (tenkit/lib/python/tenkit/supernova/alerts.py; C++: lib/assembly/src/10X/Martian.h:12-111).  Martian::exit(msg) writes
# {"exit": [msg], ...} to <out>/martian_alerts.json and leaves with code 185 (cMagic); alarms go to the same file and to
# <out>/alerts_rollup.txt.  The stage writes <out>/alerts.list first (thresholds DF checks its metrics against).
class StageExit(Exception):
    """martian.exit(msg): the stage ends with a message for the user (not a stack trace)."""


class _PrintHandlers:
    """What `import martian` offers inside a Martian py stage; used when that module is not there (tests, plain runs)."""

    def __init__(self):
        self.posted = []

    def alarm(self, m):
        self.posted.append(("alarm", m)); print("[alarm]", m)

    def log_info(self, m):
        self.posted.append(("log_info", m)); print("[log_info]", m)

    def log_warn(self, m):
        self.posted.append(("log_warn", m)); print("[log_warn]", m)

    def throw(self, m):
        raise Exception(m)

    def exit(self, m):
        raise StageExit(m)


def _martian_handlers():
    try:
        import martian  # the real adapter, inside a Martian run
        if hasattr(martian, "exit") and hasattr(martian, "alarm"):
            return martian
    except Exception:
        pass
    return _PrintHandlers()


def load_alerts(alarms_json: str) -> dict:
    """alerts.py:31-42 (check_alert :17-29): {stage: [{action, metric, compare, threshold, message}, ...]}."""
    import json
    alerts = json.loads(Path(alarms_json).read_text())
    for stage, lst in alerts.items():
        for a in lst:
            for key in ("action", "metric", "compare", "threshold", "message"):
                if key not in a:
                    raise Exception("incorrectly formatted alert, see stdout.")
            if a["compare"] not in ("<", ">"):
                raise Exception("invalid value for compare in alert")
            if type(a["threshold"]) not in (int, float):
                raise Exception("%s: invalid type for threshold" % type(a["threshold"]))
    return alerts


def write_stage_alerts(stage: str, path: str, alarms_json: str, alerts_file: str = "alerts.list") -> str:
    """alerts.py:44-61: the stage's alerts as the line-oriented list the C++ side reads."""
    alerts = load_alerts(alarms_json)
    os.makedirs(path, exist_ok=True)
    if stage not in alerts:
        raise Exception("No alerts found for stage %s" % stage)
    out = os.path.join(path, alerts_file)
    with open(out, "w") as f:
        for a in alerts[stage]:
            f.write("#\n" + a["metric"] + "\n" + str(a["threshold"]) + "\n" + a["compare"] + "\n" + a["action"] + "\n" + a["message"] + "\n")
    return out


def find_alarms_json():
    """$SNK_ALARMS_JSON, or the pipeline's own tenkit/alarms/alarms-supernova.json when tenkit is importable."""
    p = os.environ.get("SNK_ALARMS_JSON")
    if p:
        return p
    try:
        import tenkit.constants as tc
        q = os.path.join(tc.ALARMS_LOCATION, "alarms-supernova.json")
        return q if os.path.exists(q) else None
    except Exception:
        return None


class SupernovaAlarms:
    """alerts.py:82-161."""
    SN_STAGE_ALARMS, SN_ROLLUP_ALARMS = "martian_alerts.json", "alerts_rollup.txt"
    SN_ALARM_HEAD = "The following warning(s) were issued prior to encountering an error:"
    SN_UNEXPECTED_TEXT = "An unexpected error has occurred."

    def __init__(self, base_dir, handlers=None, delete=True):
        self._alarms_file = os.path.join(base_dir, self.SN_STAGE_ALARMS)
        self._rollup_file = os.path.join(base_dir, self.SN_ROLLUP_ALARMS)
        self.h = handlers or _martian_handlers()
        self._posted = False
        if delete:
            self.check_delete()

    def exit(self, msg=None):
        full = self.SN_UNEXPECTED_TEXT if msg is None else msg
        if os.path.exists(self._rollup_file):
            issued = list(set(open(self._rollup_file).read().split("\n")))      # a restarted stage repeats its alarms
            full += "\n\n" + self.SN_ALARM_HEAD + "\n\n" + "\n".join(issued) + "\n"
        self.h.exit(full)

    def post(self):
        import json
        self._posted = True
        if not os.path.exists(self._alarms_file):
            return
        alerts = json.loads(open(self._alarms_file).read())
        handlers = {"alarm": self.h.alarm, "log_info": self.h.log_info, "log_warn": self.h.log_warn, "throw": self.h.throw}
        meta, exit_str = [], ""
        for k, v in alerts.items():
            if k == "exit":
                exit_str = ";".join(v)
            elif k not in handlers:
                meta.append("unknown key {} in {} (BUG)".format(k, self._alarms_file))
            else:
                for post in v:
                    handlers[k](post)
        for m in meta:
            self.h.alarm(m)
        if exit_str:
            self.exit(exit_str)

    def check_delete(self):
        if os.path.isfile(self._alarms_file):
            os.unlink(self._alarms_file)


def build_df_command(args, out_dir: str, mspedges: str | None) -> list[str]:
    """The argv of df/__init__.py:123-139 (select_frac is pinned to 1.0 there, :121)."""
    cmd = ["DF", "LR_SELECT_FRAC={:.8f}".format(1.0), "LR=" + _get(args, "reads"), "OUT_DIR=" + out_dir,
           "MAX_MEM_GB=" + str(_get(args, "__mem_gb")), "NUM_THREADS=" + str(_get(args, "__threads"))]
    if mspedges is not None:
        cmd.append("MSPEDGES={}".format(mspedges))
    if _get(args, "pipeline_id") is not None:
        cmd.append("PIPELINE={}".format(_get(args, "pipeline_id")))
    if _get(args, "known_sample_id") is not None:
        cmd.append("SAMPLE={}".format(_get(args, "known_sample_id")))
    addin = _get(args, "addin")
    if addin is not None and "DF" in addin:
        cmd.extend(addin["DF"].split())
    if _get(args, "nodebugmem") is None or not _get(args, "nodebugmem"):
        cmd.append("TRACK_SOME_MEMORY=True")
    return cmd


# ReadDataType, 10X/DfTools.h:23-28
_UNBAR_10X, _BAR_10X = 2, 3


def read_dti(path) -> list[tuple[int, int]]:
    """<head>.dti = BINWRITE vec<DataSet>; DataSet = {ReadDataType dt (u8, padded to 8); int64 start}, trivially
    serialisable (10X/DfTools.h:30-46).  Returns [(dt, start), ...]."""
    raw = Path(path).read_bytes()
    if len(raw) < 16 or raw[:8] != b"BINWRITE":
        raise Exception("not a BINWRITE file: " + str(path))
    (n,) = struct.unpack_from("<Q", raw, 8)
    if 16 + 16 * n > len(raw):
        raise Exception("truncated dataset index: " + str(path))
    return [(raw[16 + 16 * i], struct.unpack_from("<q", raw, 24 + 16 * i)[0]) for i in range(n)]


def bc_start_of(head: str) -> int:
    """Index of the first read of a 10X datatype: reads below it ignore the barcode rule (10X/DF.cc:358-363 ->
    buildReadQGraph48's ignBcBelow, RunStages.cc:398).  No .dti next to the reads (a bare fastb/qualp/bci triple, as the
    tests use): every read is barcoded data, start 0."""
    dti = head + ".dti"
    if not os.path.exists(dti):
        return 0
    for dt, start in read_dti(dti):
        if dt in (_UNBAR_10X, _BAR_10X):          # "R data must come first"
            return int(start)
    return 0


def graph_params_of(args) -> dict:
    """K / MIN_FREQ / MIN_BC / MIN_QUAL as the stock DF would see them: its defaults (10X/DF.cc:138-141) overridden by
    KEY=VALUE words of addin["DF"] (df/__init__.py:135-136) -- the GPU-built graph must be the one DF would have built."""
    out = {"K": 48, "MIN_FREQ": 3, "MIN_BC": 2, "MIN_QUAL": 7}
    addin = _get(args, "addin")
    if addin is not None and "DF" in addin:
        for word in addin["DF"].split():
            k, _, v = word.partition("=")
            if k in out:
                out[k] = int(v)
    return out


def load_stage_inputs(reads: str, quals: str, bci: str):
    """fastb/qualp/bci -> (rows, lens, quals, bc, read_len) through the one-thread HOST readers; bc = barcode ordinal per read
    (10X/DF.cc:464-469).  The stage itself does not come through here any more (compute_mspedges decodes on the device): this is the
    byte-for-byte check of that path and a convenience for small inputs."""
    from . import formats
    rows, lens, mx = formats.read_fastb(reads)
    n = rows.shape[0]
    q = formats.read_qualp(quals, n, max(mx, 1))
    bc, _ = formats.read_bci(bci, n)
    return rows, lens, q, bc, mx


def compute_mspedges(args, out_dir: str, device: int = 0, bc_start: int | None = None, stats: dict | None = None) -> str:
    """Unitigs of the stage inputs on the GPU -> <out_dir>/asm_graph.bv (what _ASM_SN.asm_graph would have supplied).  The three files are
    decoded ON THE DEVICE (snk_df_open / snk_dev_ingest_df_count_graph, include/snk.h): raw byte ranges of a slab of reads go up, kernels
    make rows / quality rows / barcode ids of them and the slab is partitioned while the next one is being read -- the reference's
    bases.ReadAll + VirtualMasterVec<PQVec> + barcode index expansion (10X/DF.cc:265-272,345,464-469,595-597)."""
    import ctypes as C
    from . import lib as _lib
    reads = _get(args, "reads")
    if bc_start is None:
        bc_start = bc_start_of(reads[:-len(".fastb")])
    gp = graph_params_of(args)
    lib = _lib.load()
    err = C.create_string_buffer(512)

    def ok(rc):
        if rc:
            raise _lib.SnkError(rc, err.value.decode(errors="replace"))

    h, files = C.c_void_p(), C.c_void_p()
    ok(lib.snk_ctx_create(device, C.byref(h), err, 512))
    try:
        info = _lib.SnkDfInfo()
        ok(lib.snk_df_open(reads.encode(), _get(args, "quals").encode(), _get(args, "bci").encode(), C.byref(files), C.byref(info), err, 512))
        p = _lib.SnkParams()
        p.K, p.min_qual, p.min_freq, p.min_bc = gp["K"], gp["MIN_QUAL"], gp["MIN_FREQ"], gp["MIN_BC"]
        p.flags = 2                                    # SNK_F_UNSORTED_TABLE: the hand-off is the unitigs
        res, st = _lib.SnkDevResult(), _lib.SnkDevIngest()
        threads = int(_get(args, "__threads") or 0)
        # "barcoded datatypes start at" (RunStages.cc:398, DF.cc:358-363) = ign_bc_below
        ok(lib.snk_dev_ingest_df_count_graph(h, files, 0, info.n_reads, 0, min(threads, 64), 0, C.byref(p), bc_start, C.byref(res), C.byref(st), err, 512))
        d_img, nb = C.c_void_p(0), C.c_uint64(0)
        ok(lib.snk_dev_bv_image(h, gp["K"], res.n_unitigs, res.unitig_off, res.unitig_bases, 1, C.byref(d_img), C.byref(nb), None, err, 512))
        image = np.empty(int(nb.value), dtype=np.uint8)
        _lib.check(lib.snk_dev_download(h, d_img, image.ctypes.data, image.nbytes, None))
        if stats is not None:
            stats.update(n_reads=int(st.n_reads), file_bytes=int(st.text_bytes), ingest_seconds=float(st.seconds), n_slabs=int(st.n_batches),
                         n_unitigs=int(res.n_unitigs), n_kmers=int(res.n_kmers), n_instances=int(res.n_instances), count_graph_ms=float(res.phase_ms[7]))
    finally:
        if files:
            lib.snk_df_close(files)
        lib.snk_ctx_destroy(h)
    os.makedirs(out_dir, exist_ok=True)
    path = str(Path(out_dir) / "asm_graph.bv")
    with open(path, "wb") as f:
        f.write(image.tobytes())
    return path


def main(args, outs, run_df: bool = True, handlers=None):
    print("__threads=", _get(args, "__threads"))
    print("__mem_gb=", _get(args, "__mem_gb"))
    h1 = check_exclude(_get(args, "reads"), ".fastb")
    h2 = check_exclude(_get(args, "quals"), ".qualp")
    h3 = check_exclude(_get(args, "bci"), ".bci")
    if h1 != h2 or h2 != h3:
        raise Exception("something wrong with filenames passed in")
    out_dir = _get(outs, "default")
    mspedges = _get(args, "mspedges")
    if mspedges is None:
        mspedges = compute_mspedges(args, out_dir, device=int(os.environ.get("SNK_DEVICE", "0")))
    cmd = build_df_command(args, out_dir, mspedges)
    print(" ".join(cmd))
    if run_df:
        df_bin = os.environ.get("SNK_DF_BIN") or shutil.which("DF")
        if df_bin is None:
            raise RuntimeError("the stock DF binary is not on PATH (set SNK_DF_BIN); unitigs are in " + mspedges)
        # alerts.list for the C++ side, then DF; its alarms / Martian::exit (exit code 185) come back through files
        # (df/__init__.py:139-166)
        aj = find_alarms_json()
        if aj:
            write_stage_alerts("df", out_dir, aj)
        bell = SupernovaAlarms(out_dir, handlers=handlers)
        try:
            subprocess.check_call([df_bin] + cmd[1:])
        except subprocess.CalledProcessError as e:
            bell.post()                                          # a Martian::exit of the C++ code leaves from here
            bell.exit(process_return_code(e.returncode))         # anything else: signal / out of memory / unexpected
        bell.post()
        for f in glob.glob("*.mm"):
            shutil.move(f, os.path.join(out_dir, "stats"))
    return cmd
