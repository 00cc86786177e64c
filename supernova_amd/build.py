"""Build libsnk.so (HIP/gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles gfx950 without a GPU, so this runs on the CPU-only build container as well as
on the MI355X box.  The shared object is written next to this file (supernova_amd/libsnk.so): it is
git-ignored but travels with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "csrc" / "_obj"
LIB = HERE / "libsnk.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
    "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-ffp-contract=off",
]


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _newest_header() -> float:
    hs = list(CSRC.glob("*.h")) + [HERE.parent / "include" / "snk.h"]
    return max(h.stat().st_mtime for h in hs)


def build(force: bool = False, verbose: bool = True) -> Path:
    OBJ.mkdir(exist_ok=True)
    hdr = _newest_header()
    todo = []
    objs = []
    for src in _sources():
        obj = OBJ / (src.stem + ".o")
        objs.append(obj)
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr):
            todo.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)

    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(cc, todo))
    if todo or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        # The HIP runtime is NOT a DT_NEEDED of libsnk.so: the host process supplies it (torch's bundled
        # libamdhip64 when the host is Python/torch, /opt/rocm's otherwise -- see supernova_amd/lib.py and
        # INTEGRATION.md).  Two HIP runtimes in one process do not work.
        cmd = [os.environ.get("CXX", "g++"), "-shared", "-fPIC", "-o", str(LIB), *map(str, objs), "-lz", "-ldl", "-lpthread"]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    _build_host(verbose)
    return LIB


BIN = HERE / "bin"


def _build_host(verbose: bool) -> None:
    """C++ host programs above the C ABI (plain g++): here the process owns the HIP runtime, so they link /opt/rocm's."""
    BIN.mkdir(exist_ok=True)
    rocm_lib = Path(os.environ.get("ROCM_PATH", "/opt/rocm")) / "lib"
    for src in sorted((CSRC / "host").glob("*.cc")):
        exe = BIN / src.stem
        if exe.exists() and exe.stat().st_mtime >= max(src.stat().st_mtime, LIB.stat().st_mtime):
            continue
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", f"-I{rocm_lib.parent / 'include'}", str(src), "-o", str(exe), f"-L{HERE}", "-lsnk",
               f"-L{rocm_lib}", "-lamdhip64", f"-Wl,-rpath,{HERE}", f"-Wl,-rpath,{rocm_lib}", "-Wl,-rpath,$ORIGIN/.."]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"host program {src.name} failed to build:\n{r.stdout}\n{r.stderr}")


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
