"""Build libfgcn.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m fusion_gcn_amd.build          # compile if sources are newer than the library
    python -m fusion_gcn_amd.build --force

hipcc cross-compiles gfx950 code objects without a GPU.  Objects go to fusion_gcn_amd/_build/, the library to
fusion_gcn_amd/libfgcn.so (git-ignored, shipped to the GPU box with the tree).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libfgcn.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "fgcn.h"))
    return hdrs


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    lib_m = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > lib_m for f in sources() + _deps())


def build(force: bool = False, verbose: bool = False, out_dir: str = None) -> str:
    """Compile every csrc/*.hip for gfx950 and link libfgcn.so.  Default: incremental, in-tree.  ``out_dir``: a from-scratch build
    of all sources into that directory (objects and library), leaving the in-tree library alone -- what a fresh clone does
    (tests/test_abi.py::test_fresh_build_from_sources)."""
    obj_dir, lib = (OBJ, LIB) if out_dir is None else (out_dir, os.path.join(out_dir, "libfgcn.so"))
    force = force or out_dir is not None
    if not force and not needs_build():
        return LIB
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    dep_m = max(os.path.getmtime(f) for f in _deps())

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), dep_m):
            return obj
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, sources()))
    tmp = lib + ".tmp"
    r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
