"""A/B: identity-shortcut gradients written by the BatchNorm-backward kernels (dx read-modify-write twice) vs added by joint_dagg
from the sign images (fgcn_joint_dagg extra1/extra2), at the three widths of the model, B = 128 samples."""
import sys

import torch

sys.path.insert(0, ".")
from fusion_gcn_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def t_ms(fn, reps=10):
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
    r = lambda *s: torch.randn(*s, device=dev)      # noqa: E731
    x, u, y, d_o, dg, dagg = r(B, T, V, C), r(B, T, V, C), r(B, T, V, C), r(B, T, V, C), r(B, T, V, C), r(B, T, V, 3 * C)
    a_hat = r(B, 3, V, V) * 0.2
    vec = torch.stack([torch.zeros(C), torch.ones(C), torch.ones(C), torch.zeros(C)]).to(dev).contiguous()
    o, o_sign = ops.bn_act(u, vec, x, None, relu=True, sign_mask=True)
    g, g_sign = ops.bn_act(y, vec, x, None, relu=True, sign_mask=True)
    dx = torch.empty_like(x)

    def old_path():
        ops.bn_act_bwd(d_o, None, u, vec, x, None, res_mode=1, train=True, db=dx, sign_mask=o_sign)
        ops.bn_act_bwd(dg, None, y, vec, x, None, res_mode=1, train=True, db=dx, db_accumulate=True, sign_mask=g_sign)
        ops.joint_dagg(x, dagg, a_hat, dx, accumulate=True)

    def new_path():
        ops.bn_act_bwd(d_o, None, u, vec, x, None, res_mode=1, train=True, need_db=False, sign_mask=o_sign)
        ops.bn_act_bwd(dg, None, y, vec, x, None, res_mode=1, train=True, need_db=False, sign_mask=g_sign)
        ops.joint_dagg(x, dagg, a_hat, dx, accumulate=False, gated=[(d_o, o_sign), (dg, g_sign)])

    def dagg_plain():
        ops.joint_dagg(x, dagg, a_hat, dx, accumulate=True)

    def dagg_gated():
        ops.joint_dagg(x, dagg, a_hat, dx, accumulate=False, gated=[(d_o, o_sign), (dg, g_sign)])

    old_path()
    want = dx.clone()
    new_path()
    err = float((dx - want).norm() / want.norm())
    res = [t_ms(old_path), t_ms(dagg_plain)]
    for three in (0, 8):
        _lib.load().fgcn_set_tuning(6, three)
        res += [t_ms(new_path), t_ms(dagg_gated)]
    _lib.load().fgcn_set_tuning(6, 0)
    print(f"C={C} T={T}: old chain {res[0]:.3f} ms (joint_dagg {res[1]:.3f}) | gated, 2 WG/CU: chain {res[2]:.3f} (joint_dagg {res[3]:.3f}) | "
          f"gated, 3 WG/CU: chain {res[4]:.3f} (joint_dagg {res[5]:.3f}) | rel diff {err:.1e}")
