#!/usr/bin/env python3
"""Per-kernel compiler report for one HIP source of libfgcn (no GPU needed): registers, LDS, scratch, occupancy from
hipcc's kernel-resource-usage remarks, plus instruction counts from the gfx950 assembly (MFMA, buffer/global/LDS
memory instructions, full `vmcnt(0)` drains, exec-mask regions, branches).

    python tools/kres.py joint [name-filter]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include"]


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()


def main():
    stem = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    src = os.path.join(ROOT, "fusion_gcn_amd", "csrc", f"fgcn_{stem}.hip")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", f"{tmp}/o.o"],
                           capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        res = {}
        pat = (r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
               r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)")
        for m in re.finditer(pat, r.stderr, re.S):
            res[m.group(1)] = m.groups()[1:]
        r = subprocess.run([HIPCC, *FLAGS, "-S", "--cuda-device-only", src, "-o", f"{tmp}/o.s"],
                           capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        lines = open(f"{tmp}/o.s").read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for si in starts:
        mangled = lines[si].split(":")[0]
        name = demangle(mangled)
        if flt and flt not in name:
            continue
        ei = next(i for i in range(si, len(lines)) if lines[i].startswith(".Lfunc_end"))    # (the first s_endpgm may be an early exit)
        body = [l for l in lines[si:ei] if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
        cnt = lambda p: sum(1 for l in body if re.search(p, l))
        v, a, scr, occ, lds = res.get(mangled, ("?",) * 5)
        print(f"{name[:76]:76s} vgpr={v:>3} agpr={a:>3} scratch={scr:>3} occ={occ} lds={lds:>6} | instr={len(body):5d} "
              f"mfma={cnt('v_mfma'):3d} buf_ld={cnt('buffer_load'):3d} buf_st={cnt('buffer_store'):3d} "
              f"g_ld={cnt('global_load'):3d} g_st={cnt('global_store'):3d} ds={cnt(r'ds_(read|write)'):3d} "
              f"vmcnt0={cnt(r'vmcnt.0.'):3d} saveexec={cnt('saveexec'):3d} branch={cnt('s_cbranch'):3d}")


if __name__ == "__main__":
    main()
