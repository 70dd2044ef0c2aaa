#!/usr/bin/env python3
"""Timing-probe builds of libfgcn: ONE source recompiled with extra -D flags, linked with the tree's other objects.

    python tools/build_probe.py <name> <source.hip> -DFGCN_PROBE_PW=1 [...]     ->  tools/probes/libfgcn_<name>.so

Run a tool against it with FGCN_LIB=$PWD/tools/probes/libfgcn_<name>.so (fusion_gcn_amd/_lib.py).  The in-tree library is built
first (incrementally) so that the other objects are current.  Probe libraries are git-ignored and travel to the GPU box: delete
tools/probes/libfgcn_*.so and tools/probes/_obj/ when the measurement is done (every snapshot ships them otherwise).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fusion_gcn_amd import build as B  # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build()
    src_path = os.path.join(B.CSRC, os.path.basename(src))
    out_dir = os.path.join(ROOT, "tools", "probes", "_obj")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, f"{name}_{os.path.basename(src)[:-4]}.o")
    subprocess.run([B._hipcc(), *B.FLAGS, *flags, "-c", src_path, "-o", obj], check=True)
    others = [os.path.join(B.OBJ, os.path.basename(s)[:-4] + ".o") for s in B.sources() if os.path.basename(s) != os.path.basename(src)]
    lib = os.path.join(ROOT, "tools", "probes", f"libfgcn_{name}.so")
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib, obj, *others], check=True)
    print(lib)


if __name__ == "__main__":
    main()
