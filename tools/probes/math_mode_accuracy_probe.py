"""accuracy of the temporal conv in the three math modes against float64"""
import sys, torch
sys.path.insert(0, ".")
from fusion_gcn_amd import ops, block
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T, V, c = 8, 60, 25, 128
g = torch.randn(B, T, V, c, device=dev)
wt = torch.randn(9, c, c, device=dev) * (9 * c) ** -0.5
W = {"t": wt, "t_t": wt.permute(0, 2, 1).contiguous()}
bias = torch.randn(c, device=dev)
# float64 reference
gd = g.double().cpu(); wd = wt.double().cpu()
ref = torch.zeros(B, T, V, c, dtype=torch.float64)
for j in range(9):
    d = j - 4
    lo, hi = max(0, -d), min(T, T - d)
    ref[:, lo:hi] += gd[:, lo + d:hi + d] @ wd[j]
ref += bias.double().cpu()
for mode in ("f32", "bf16", "bf16x3"):
    with ops.math_mode(mode):
        W["t4"] = ops.pack_conv(wt)
        u = torch.empty(B, T, V, c, device=dev)
        block.temporal_fwd(g, u, W, bias, 9, 1, stats=True)
        e = (u.double().cpu() - ref)
        print(f"{mode:7s} rel-L2 {float(e.norm() / ref.norm()):.3e}  max-abs {float(e.abs().max()):.3e}  (|ref| max {float(ref.abs().max()):.2f})")
