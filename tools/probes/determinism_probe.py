"""Probe: is the AGCN step bitwise reproducible?  Runs fwd+bwd of the headline model N times on the same batch in one process and
compares every parameter gradient with the first run's, bit for bit (run two copies at once to add contention:
`python tools/probes/determinism_probe.py 8 12 & python tools/probes/determinism_probe.py 8 12; wait`)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench  # noqa: E402
from fusion_gcn_amd import ops  # noqa: E402
from fusion_gcn_amd.loss import cross_entropy  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
dev = torch.device("cuda:0")
ops.set_math_mode(mode)
model = bench.build_model(dev)
g = torch.Generator().manual_seed(1)
S = bench.SHAPE
x = torch.randn(clips, S["M"], S["T"], S["V"], S["C"], generator=g).to(dev)
y = torch.randint(0, S["classes"], (clips,), generator=g).to(dev)
names = [n for n, _ in model.named_parameters()]
first = None
bad = {}
for r in range(runs):
    for p in model.parameters():
        p.grad = None
    loss = cross_entropy(model(x), y)
    loss.backward()
    torch.cuda.synchronize()
    grads = [p.grad.detach().clone() for p in model.parameters()]
    if first is None:
        first, loss0 = grads, float(loss)
        continue
    for n, a, b in zip(names, first, grads):
        if not torch.equal(a, b):
            bad[n] = max(bad.get(n, 0.0), float((a - b).abs().max() / (a.abs().max() + 1e-30)))
    if float(loss) != loss0:
        bad["<loss>"] = abs(float(loss) - loss0)
print(f"pid {os.getpid()} clips {clips} runs {runs} mode {mode}: {len(bad)} of {len(names)} gradients differ between runs")
for n, e in sorted(bad.items(), key=lambda t: -t[1])[:25]:
    print(f"   {n:48s} {e:.2e}")
