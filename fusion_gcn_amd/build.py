"""Build libfgcn.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m fusion_gcn_amd.build          # compile if sources are newer than the library
    python -m fusion_gcn_amd.build --force
    python -m fusion_gcn_amd.build --host-asan   # the launchers' HOST code under AddressSanitizer + UBSan (CPU box; see build_host_asan)

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


ASAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fsanitize=address,undefined",
              "-fno-gpu-sanitize", "-Wno-unused-function", "-Wno-pass-failed"]


def asan_runtime() -> str:
    """The shared AddressSanitizer runtime of ROCm's clang (to LD_PRELOAD into a Python that loads the instrumented library)."""
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not hits:
        raise RuntimeError("libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm")
    return hits[-1]


def build_host_asan(verbose: bool = False) -> str:
    """SURVEY.md section 5 (sanitizers), CPU side only: every csrc/*.hip compiled with -fsanitize=address,undefined on the HOST half
    (-fno-gpu-sanitize: the gfx950 code objects are built as usual -- GPU sanitizers are not available on this pool) into
    fusion_gcn_amd/_build/asan/libfgcn_asan.so.  What it instruments is everything a call executes before its first launch: argument
    validation, tile geometry, FastDiv tables, workspace / slab sizes, descriptor packing, the per-thread contexts.  Load it with
    FGCN_LIB=<path> and LD_PRELOAD=asan_runtime() (tests/test_abi.py::test_host_code_under_sanitizers does).  Incremental."""
    obj_dir = os.path.join(OBJ, "asan")
    lib = os.path.join(obj_dir, "libfgcn_asan.so")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    dep_m = max(os.path.getmtime(f) for f in _deps())

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), dep_m):
            return obj, False
        cmd = [hipcc, *ASAN_FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (host asan) failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj, True

    with ThreadPoolExecutor(max_workers=6) as ex:
        done = list(ex.map(compile_one, sources()))
    if any(fresh for _, fresh in done) or not os.path.exists(lib):
        tmp = lib + ".tmp"
        r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fsanitize=address,undefined", "-shared-libsan",
                            "-o", tmp, *[o for o, _ in done]], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link (host asan) failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, lib)
    return lib


if __name__ == "__main__":
    if "--host-asan" in sys.argv:
        print(build_host_asan(verbose=True))
        print("LD_PRELOAD=" + asan_runtime())
    else:
        print(build(force="--force" in sys.argv, verbose=True))
