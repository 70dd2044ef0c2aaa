import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from conftest import rel_l2
from oracle import filler
from fusion_gcn_amd import ops
from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
from fusion_gcn_amd.models.mmargcn.agcn import Model
from fusion_gcn_amd.util import Graph
ref = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "model.npz"))
dev = torch.device("cuda:0")
shape, classes = (2, 2, 32, 25, 3), 60
model = Model(shape[1:], classes, Graph(ntu.skeleton_edges, center_joint=ntu.center_joint))
filler.fill_state_dict(model.state_dict())
model = model.to(dev).eval()
x = torch.from_numpy(filler.skeleton_input("x.cfg2_small", shape, empty_second_body=True)).float().to(dev)
want = ref["cfg2_small.eval.logits"]
res = {}
for mode in ("f32", "bf16x3", "bf16"):
    with ops.math_mode(mode), torch.no_grad():
        for name, opts in (("two-pass", "fused_inference=0"), ("fused", ""), ("fused, spatial two-pass", "fused_inference=1")):
            with ops.context() as c:
                c.paths.update_from(opts)
                out = model(x)
            res[(mode, name)] = out
            print(f"{mode:7s} {name:26s} vs reference {rel_l2(out.cpu().numpy(), want):.3e}   vs two-pass {float((out - res[(mode, 'two-pass')]).norm() / res[(mode, 'two-pass')].norm()):.3e}")
# which half matters in bf16: monkeypatch one fused kernel off at a time
with ops.math_mode("bf16"), torch.no_grad():
    conv, sp = ops.tconv_halo_bn_relu, ops.spatial_fwd_tile_bn_relu
    import fusion_gcn_amd.block as blk
    avail = ops.spatial_fwd_tile_available
    ops.spatial_fwd_tile_available = lambda *a: False
    out = model(x)
    ops.spatial_fwd_tile_available = avail
    print("bf16 fused temporal only        vs reference", f"{rel_l2(out.cpu().numpy(), want):.3e}")
