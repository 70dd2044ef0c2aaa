"""What would config 5 (math mode bf16) lose if EVERY activation-sized tensor were stored as bfloat16 (the reference's autocast semantics)
instead of only the ones bf16 MFMA staging reads?  Emulation on the present kernels: the float32 outputs of the block's kernels are
rounded to bfloat16 in place right after the launch that writes them (Y, U, O / x, the shortcut convs, dG, dx), BatchNorm sums stay
the producers' (from the unrounded accumulators) -- exactly what kernels with bfloat16 stores would do.  Against the reference's
logits / loss (tests/golden/model.npz) and the same model's bf16x3-mode gradient.   python tools/probes/half_act_probe.py [T]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn.functional as F
from conftest import rel_l2
from oracle import filler
from fusion_gcn_amd import ops
from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
from fusion_gcn_amd.models.mmargcn.agcn import Model
from fusion_gcn_amd.util import Graph

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ref = np.load(os.path.join(ROOT, "tests", "golden", "model.npz"))
dev = torch.device("cuda:0")
shape, classes = (2, 2, T, 25, 3), 60
model = Model(shape[1:], classes, Graph(ntu.skeleton_edges, center_joint=ntu.center_joint))
filler.fill_state_dict(model.state_dict())
model = model.to(dev).train()
x = torch.from_numpy(filler.skeleton_input("x.cfg2_small", shape, empty_second_body=True)).float().to(dev)
y = torch.from_numpy(ref["cfg2_small.labels"]).to(dev)

WHICH = set()


def rnd_(t):
    if t is not None and t.dtype == torch.float32 and t.dim() == 4 and t.shape[1] * t.shape[2] > 1:
        t.copy_(t.bfloat16().float())


def wrap_out(name, tag, idx=None, key=None, ret=None):
    f = getattr(ops, name)

    def g(*a, **k):
        r = f(*a, **k)
        if tag in WHICH:
            if idx is not None:
                rnd_(a[idx] if len(a) > idx else k.get(key))
            if ret is not None:
                rnd_(r[ret] if isinstance(r, tuple) else r)
        return r
    setattr(ops, name, g)


wrap_out("spatial_fwd_tile", "y", ret=0)
wrap_out("spatial_fwd", "y", ret=0)
wrap_out("tconv_halo", "u", idx=2, key="out")          # U forward, dG backward (both float32 outputs of the halo conv)
wrap_out("rows_gemm", "u", idx=2, key="out")
wrap_out("pw_gemm", "u", idx=2, key="out")
wrap_out("bn_act", "o", ret=0)
wrap_out("data_bn_apply", "o", ret=0)
wrap_out("spatial_bwd_tile", "dx", idx=4, key="dx")
wrap_out("emb_dx_tile", "dx", idx=3, key="dx")
wrap_out("joint_dagg", "dx", idx=3, key="dx")


def run(mode, which):
    WHICH.clear()
    WHICH.update(which)
    model.zero_grad(set_to_none=True)
    with ops.math_mode(mode):
        logits = model(x)
        loss = F.cross_entropy(logits, y)
        loss.backward()
    g = torch.cat([p.grad.detach().double().flatten() for _, p in model.named_parameters()])
    return logits.detach().double(), float(loss), g


l0, loss0, g0 = run("bf16x3", ())
print(f"T = {T}; bf16x3: logits vs the reference {rel_l2(l0.cpu().numpy(), ref['cfg2_small.train.logits']) if T == 32 else float('nan'):.2e}")
for which in ((), ("y",), ("u",), ("y", "u"), ("o",), ("y", "u", "o"), ("dx",), ("y", "u", "o", "dx")):
    l, loss, g = run("bf16", which)
    cos = float(torch.dot(g, g0) / (g.norm() * g0.norm()))
    print(f"bf16 + bfloat16 storage of {'/'.join(which) or 'nothing more':14s}: logits vs bf16x3 {float((l - l0).norm() / l0.norm()):.2e}"
          + (f", vs the reference {rel_l2(l.cpu().numpy(), ref['cfg2_small.train.logits']):.2e}" if T == 32 else "")
          + f", |loss - bf16x3| {abs(loss - loss0):.2e}, gradient cosine {cos:.4f}, rel-L2 {float((g - g0).norm() / g0.norm()):.2e}")
