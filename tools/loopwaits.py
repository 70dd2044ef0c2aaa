#!/usr/bin/env python3
"""Static check of a libfgcn source for full `s_waitcnt vmcnt(0)` drains INSIDE loops (no GPU needed):
    python tools/loopwaits.py twgrad [name-filter]
Per kernel: every loop (a backward branch to an earlier label) with its instruction count, MFMAs, buffer loads / stores and the
number of vmcnt(0) waits inside it.  A drain inside a loop that also stores means every iteration waits for its write acknowledgements
(vmcnt counts loads and stores in issue order): the pattern behind the epilogue fixes of round 3."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stem, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
src = os.path.join(ROOT, "fusion_gcn_amd", "csrc", f"fgcn_{stem}.hip")
with tempfile.TemporaryDirectory() as tmp:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include", "-S", "--cuda-device-only",
                    src, "-o", f"{tmp}/o.s"], check=True, capture_output=True)
    lines = open(f"{tmp}/o.s").read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)] + [len(lines)]
for a, b in zip(starts, starts[1:]):
    name = subprocess.run(["c++filt", lines[a].split(":")[0]], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    body = lines[a:b]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t < i:
                loops.append((t, i))
    out = []
    for t, i in loops:
        seg = [x for x in body[t:i] if x.startswith("\t") and not x.startswith("\t.") and not x.startswith("\t;")]
        c = lambda p: sum(1 for x in seg if re.search(p, x))
        if c(r"vmcnt\(0\)"):
            out.append(f"    loop {len(seg):5d} instr  mfma={c('v_mfma'):3d} ld={c('buffer_load|global_load'):3d} st={c('buffer_store|global_store'):3d} "
                       f"vmcnt0={c(r'vmcnt.0.'):2d} barriers={c('s_barrier')}")
    if out:
        print(name[:110])
        print("\n".join(out))
