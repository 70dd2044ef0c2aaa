"""Probe: where do the step's small device-to-device copies come from?  (profiles/r02: 141 `__amd_rocclr_copyBuffer` launches per
step.)  Runs eager fwd+bwd steps of the headline model at a few clips under torch.profiler with Python stacks and prints, per
source line, how many aten::copy_ / aten::clone / aten::contiguous / aten::_to_copy calls it issued per step."""
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from fusion_gcn_amd import ops  # noqa: E402
from fusion_gcn_amd.loss import cross_entropy  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
ops.set_math_mode("bf16x3")
model = bench.build_model(dev)
g = torch.Generator().manual_seed(1)
S = bench.SHAPE
x = torch.randn(clips, S["M"], S["T"], S["V"], S["C"], generator=g).to(dev)
y = torch.randint(0, S["classes"], (clips,), generator=g).to(dev)


def step():
    for p in model.parameters():
        p.grad = None
    for m in model.modules():
        if hasattr(m, "mark_packed_stale"):
            m.mark_packed_stale()
    cross_entropy(model(x), y).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::zeros", "aten::add_",
         "aten::add", "aten::mul", "aten::cat", "aten::empty_strided")
count = collections.Counter()
kernels = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA or "Memcpy" in ev.name or "copyBuffer" in ev.name:
        kernels[ev.name[:70]] += 1
    if ev.name in WATCH:
        where = next((f for f in ev.stack if "fusion_gcn_amd" in f or "bench.py" in f), "<autograd / torch internals>")
        count[(ev.name, where.strip()[-110:])] += 1
print("== watched aten ops per step, by first repo frame")
for (name, where), n in sorted(count.items(), key=lambda t: -t[1])[:60]:
    print(f"{n:5d}  {name:22s} {where}")
print("== device-side kernel / memcpy names per step (top 25)")
for name, n in kernels.most_common(25):
    print(f"{n:5d}  {name}")
