"""A/B of joint_dagg: f32-MFMA gram (math mode f32) vs the split-bf16 gram (bf16x3), 3 and 2 workgroups per CU; accuracy of both
against float64."""
import sys

import torch

sys.path.insert(0, ".")
from fusion_gcn_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def t_ms(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B, V = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 25
for C, T in ((64, 300), (128, 150), (256, 75)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, V, C, generator=g)
    dagg = torch.randn(B, T, V, 3 * C, generator=g)
    a_hat = torch.randn(B, 3, V, V, generator=g) * 0.2
    xs, ds = x[:4].double(), dagg[:4].double().reshape(4, T, V, 3, C)
    want_g = torch.einsum("btvc,btwkc->bkvw", xs, ds)
    xd, dd, ad = x.to(dev), dagg.to(dev), a_hat.to(dev)
    dx = torch.empty_like(xd)
    out = []
    for mode, tune in (("f32", 0), ("bf16x3", 0), ("bf16x3", 8)):
        _lib.load().fgcn_set_tuning(6, tune)
        with ops.math_mode(mode):
            part = ops.joint_dagg(xd, dd, ad, dx, accumulate=False)
            got = part.double().sum(1)[:4, :, :V, :V].cpu()
            err = float((got - want_g).norm() / want_g.norm())
            out.append((mode, tune, t_ms(lambda: ops.joint_dagg(xd, dd, ad, dx, accumulate=False)), err))
    _lib.load().fgcn_set_tuning(6, 0)
    gb = (x.numel() * 4 * (1 + 3 + 1)) / 1e9
    print(f"C={C} T={T}: " + " | ".join(f"{m} tune{t}: {ms:.3f} ms ({gb / ms:.2f} TB/s) gram err {e:.1e}" for m, t, ms, e in out))
