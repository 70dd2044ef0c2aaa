"""BASELINE config 5: the bf16-operand mode of the convolution / GEMM kernels (FGCN_MATH_BF16).

Kernel level: the result must equal the float64 product of the bf16-ROUNDED operands (same rounding, f32 accumulation),
so the tolerance stays tight.  Model level: the different contract SURVEY.md sets for bf16 (the reference under
torch.autocast(bfloat16) deviates 3.4e-3 on logits and 1.7e-1 on the flat gradient from its own f32 run): logits <= 1e-2
rel, loss <= 1e-2 abs, gradient cosine >= 0.98 against the f32 path on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import filler

pytestmark = pytest.mark.gpu
TOL = 2e-5          # f32 accumulation of exactly representable products


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def bf(x):
    """round to bfloat16 (RNE), back to float64"""
    return x.float().to(torch.bfloat16).double()


def gpu(x):
    return x.float().to(dev()).contiguous()


@pytest.fixture(autouse=True)
def _bf16_mode():
    from fusion_gcn_amd import ops
    with ops.math_mode("bf16"):
        yield
    assert ops.get_math_mode() == "f32"


def conv_ref(x, w, kt, s, T_out, bias=None):
    """x (B,T,V,K), w (kt,K,N): temporal conv, padding (kt-1)//2, stride s"""
    B, T, V, K = x.shape
    pad = (kt - 1) // 2
    out = torch.zeros(B, T_out, V, w.shape[2], dtype=torch.float64)
    for j in range(kt):
        for to in range(T_out):
            ti = to * s + j - pad
            if 0 <= ti < T:
                out[:, to] += x[:, ti] @ w[j]
    return out if bias is None else out + bias


@pytest.mark.parametrize("B,T,V,K,N,kt,s", [(2, 20, 25, 64, 64, 9, 1), (2, 21, 25, 64, 128, 9, 2), (3, 10, 18, 128, 96, 1, 1),
                                            (2, 9, 27, 4, 64, 1, 1), (1, 7, 5, 40, 36, 3, 1)])
def test_rows_gemm_bf16(B, T, V, K, N, kt, s):
    from fusion_gcn_amd import ops
    T_out = (T - 1) // s + 1
    x, w, b = rnd(B, T, V, K, seed=1), rnd(kt, K, N, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    want = conv_ref(bf(x), bf(w), kt, s, T_out, b)
    out = torch.empty(B, T_out, V, N, device=dev())
    part = ops.rows_gemm(gpu(x), gpu(w), out, K=K, N=N, tmap=ops.conv_tmap(kt, s), bias=gpu(b), stats=True)
    assert rel_l2(out.cpu().numpy(), want.numpy()) < TOL
    assert rel_l2(part.double().sum(0)[0].cpu().numpy(), want.reshape(-1, N).sum(0).numpy()) < 1e-4
    # and it is NOT the f32 result: the operands really are rounded
    assert rel_l2(out.cpu().numpy(), conv_ref(x, w, kt, s, T_out, b).numpy()) > 1e-4


@pytest.mark.parametrize("B,T,V,C,O", [(3, 20, 25, 64, 64), (2, 13, 18, 128, 256), (2, 9, 27, 64, 128)])
def test_halo_conv_bf16_forward_and_data_gradient(B, T, V, C, O):
    from fusion_gcn_amd import block, ops
    wt = rnd(9, C, O, seed=4, scale=(9 * C) ** -0.5)
    W = {"t": gpu(wt), "t_t": gpu(wt.permute(0, 2, 1))}
    W["t4"], W["t_t4"] = ops.pack_conv(W["t"]), ops.pack_conv(W["t_t"])
    x, b = rnd(B, T, V, C, seed=5), rnd(O, seed=6)
    u = torch.empty(B, T, V, O, device=dev())
    block.temporal_fwd(gpu(x), u, W, gpu(b), 9, 1, stats=True)
    assert rel_l2(u.cpu().numpy(), conv_ref(bf(x), bf(wt), 9, 1, T, b).numpy()) < TOL
    du = rnd(B, T, V, O, seed=7)
    dg = torch.empty(B, T, V, C, device=dev())
    block.temporal_dgrad(gpu(du), dg, W, 9, 1)
    want = conv_ref(bf(du), bf(wt.flip(0).permute(0, 2, 1)), 9, 1, T)          # data gradient = conv with flipped W^T
    assert rel_l2(dg.cpu().numpy(), want.numpy()) < TOL


@pytest.mark.parametrize("B,T,V,K,N,kt,s", [(3, 20, 25, 64, 64, 9, 1), (2, 21, 25, 64, 128, 9, 2), (2, 13, 18, 128, 256, 9, 1),
                                            (2, 30, 25, 192, 64, 1, 1), (1, 12, 25, 768, 256, 1, 1)])
def test_weight_gradients_bf16(B, T, V, K, N, kt, s):
    from fusion_gcn_amd import ops
    T_out = (T - 1) // s + 1
    pad = (kt - 1) // 2
    a, g = rnd(B, T, V, K, seed=8), rnd(B, T_out, V, N, seed=9)
    ab, gb = bf(a), bf(g)
    want = torch.zeros(kt, K, N, dtype=torch.float64)
    for j in range(kt):
        for to in range(T_out):
            ti = to * s + j - pad
            if 0 <= ti < T:
                want[j] += torch.einsum("bvk,bvn->kn", ab[:, ti], gb[:, to])
    if kt > 1:
        got = ops.tconv_wgrad(gpu(a), gpu(g), taps=kt, stride=s)
        assert rel_l2(got.cpu().numpy(), want.numpy()) < 1e-4
    for wide in ((False, True) if kt == 1 else (False,)):
        got = ops.rows_wgrad(gpu(a), gpu(g), K=K, N=N, tmap=ops.conv_tmap(kt, s), wide=wide)
        assert rel_l2(got.cpu().numpy(), want.numpy()) < 1e-4


@pytest.mark.parametrize("V,cin,cout", [(25, 64, 64), (18, 64, 128), (27, 128, 256), (25, 4, 64)])
def test_fused_spatial_forward_bf16(V, cin, cout):
    """Step 1 (x . A^_k) stays f32; its result and the conv_d weights are rounded for step 2."""
    from fusion_gcn_amd import ops
    B, T = 2, 6
    x, a = rnd(B, T, V, cin, seed=10), rnd(B, 3, V, V, seed=11, scale=0.3)
    wd, bias = rnd(3 * cin, cout, seed=12, scale=(3 * cin) ** -0.5), rnd(cout, seed=13)
    agg = torch.einsum("btvc,bkvw->btwkc", x.float().double(), a.float().double()).reshape(B, T, V, 3 * cin)
    want = bf(agg) @ bf(wd) + bias
    y, _ = ops.spatial_fwd(gpu(x), gpu(a), ops.pack_k4(gpu(wd).unsqueeze(0))[0], gpu(bias), Cin=cin, Cout=cout, stats=True)
    # agg is formed in f32 on the GPU and in f64 here: a handful of values round to the neighbouring bf16
    assert rel_l2(y.cpu().numpy(), want.numpy()) < 2e-3


def test_config5_model_contract():
    """cfg-2 shape at fixture size (N=2, M=2, T=32, V=25), train mode: bf16 mode against the f32 mode of the same model."""
    import torch.nn.functional as F
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.util import Graph
    shape, classes = (2, 2, 32, 25, 3), 60
    model = Model(shape[1:], classes, Graph(ntu.skeleton_edges, center_joint=ntu.center_joint))
    filler.fill_state_dict(model.state_dict())
    model = model.to(dev()).train()
    x = torch.from_numpy(filler.skeleton_input("x.cfg2_small", shape, empty_second_body=True)).float().to(dev())
    y = torch.from_numpy(filler.uniform("y.cfg2_small", (shape[0],), 0, classes).astype(np.int64)).to(dev())

    def run():
        for p in model.parameters():
            p.grad = None
        logits = model(x)
        loss = F.cross_entropy(logits, y)
        loss.backward()
        return logits.detach().clone(), float(loss.detach()), torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
    lg_b, loss_b, g_b = run()                       # bf16 (autouse fixture)
    with ops.math_mode("f32"):
        lg_f, loss_f, g_f = run()
    e_logits = float((lg_b - lg_f).norm() / lg_f.norm())
    cos = float(torch.dot(g_b, g_f) / (g_b.norm() * g_f.norm()))
    print(f"config 5: logits rel-L2 {e_logits:.2e}, loss {loss_b:.5f} vs {loss_f:.5f}, gradient cosine {cos:.4f}")
    assert e_logits < 1e-2 and abs(loss_b - loss_f) < 1e-2 and cos > 0.98
    assert e_logits > 1e-5                          # the mode really changes the arithmetic


def test_config5_model_against_the_reference(golden):
    """BASELINE config 5 pinned to the REFERENCE, not to the build's own f32 mode: bf16-mode logits / loss at the config-2 fixture
    shape against the reference's own outputs (tests/golden/model.npz ``cfg2_small.train.{logits,loss}``, written by importing
    the reference), and the flat gradient against the float64 oracle.  Contract (the reference under
    bf16 autocast deviates 3.4e-3 on logits / 1.7e-1 on the flat gradient from its own f32 run): train-mode logits <= 1e-2 rel,
    loss <= 5e-3 abs, same argmax, gradient cosine >= 0.98; eval-mode logits <= 5e-2 (running statistics do not re-normalise the
    operand-rounding error block by block as batch statistics do, so the figure is a sum of ten blocks' bf16 roundings and moves with
    every change of an f32 rounding upstream: 1.4e-2 in round 2, 2.6e-2 with round 6's two-pass eval path, 3.3e-2 with its fused
    inference kernels, 1.8e-2 with only the temporal half fused -- tools/probes/eval_paths_probe.py; the f32-class modes sit at 1e-6 ..
    1e-5 on the same fixture either way).  Measured on MI355X (r06): train logits 2.3e-3, loss 2.0e-3, cosine 0.9977, flat-gradient
    rel-L2 6.7e-2."""
    import torch.nn.functional as F
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.util import Graph
    from oracle import agcn_oracle as O
    assert ops.get_math_mode() == "bf16"
    ref = golden("model.npz")
    shape, classes = (2, 2, 32, 25, 3), 60
    model = Model(shape[1:], classes, Graph(ntu.skeleton_edges, center_joint=ntu.center_joint))
    filler.fill_state_dict(model.state_dict())
    sd = {k: (v.detach().double().clone() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    model = model.to(dev())
    x = torch.from_numpy(filler.skeleton_input("x.cfg2_small", shape, empty_second_body=True))
    y = torch.from_numpy(ref["cfg2_small.labels"])
    model.eval()
    with torch.no_grad():
        e_eval = rel_l2(model(x.float().to(dev())).cpu().numpy(), ref["cfg2_small.eval.logits"])
    model.train()
    logits = model(x.float().to(dev()))
    loss = F.cross_entropy(logits, y.to(dev()))
    loss.backward()
    e_train = rel_l2(logits.detach().cpu().numpy(), ref["cfg2_small.train.logits"])
    d_loss = abs(float(loss.detach()) - float(ref["cfg2_small.train.loss"]))
    _, _, grads_o, _ = O.loss_and_grads(x.double(), y, sd)
    flat_g = torch.cat([p.grad.detach().double().flatten().cpu() for _, p in model.named_parameters()])
    flat_o = torch.cat([grads_o[n].double().flatten() for n, _ in model.named_parameters()])
    cos = float(torch.dot(flat_g, flat_o) / (flat_g.norm() * flat_o.norm()))
    e_grad = float((flat_g - flat_o).norm() / flat_o.norm())
    print(f"config 5 vs the reference: eval logits {e_eval:.2e}, train logits {e_train:.2e}, |loss diff| {d_loss:.2e}, flat gradient vs "
          f"the fp64 oracle: cosine {cos:.4f}, rel-L2 {e_grad:.2e}")
    assert e_eval < 5e-2 and e_train < 1e-2, (e_eval, e_train)
    assert d_loss < 5e-3, d_loss
    assert cos > 0.98, cos
    assert np.array_equal(logits.detach().cpu().numpy().argmax(1), ref["cfg2_small.train.logits"].argmax(1))


@pytest.mark.parametrize("V,cin,cout", [(25, 64, 64), (18, 64, 128), (27, 128, 128)])
def test_spatial_wgrad_bf16(V, cin, cout):
    """agg is formed in f32, then agg and dy are rounded for the Cin x Cout contraction."""
    from fusion_gcn_amd import ops
    B, T = 2, 6
    x, a, dy = rnd(B, T, V, cin, seed=20), rnd(B, 3, V, V, seed=21, scale=0.3), rnd(B, T, V, cout, seed=22)
    agg = torch.einsum("btvc,bkvw->btwkc", x.float().double(), a.float().double())
    want = torch.einsum("btwkc,btwo->kco", bf(agg), bf(dy)).reshape(3 * cin, cout)
    got = ops.spatial_wgrad(gpu(x), gpu(dy), gpu(a))
    assert rel_l2(got[0].cpu().numpy(), want.numpy()) < 2e-3      # a few agg values round to the neighbouring bf16


@pytest.mark.parametrize("V,T,ic,cx,B", [(25, 13, 16, 64, 2), (25, 7, 64, 256, 2), (27, 9, 32, 64, 1), (18, 10, 32, 128, 2), (22, 31, 64, 128, 1)])
def test_embedding_backward_tile_form_bf16(V, T, ic, cx, B):
    """fgcn_emb_dx_tile / fgcn_emb_wgrad_tile with ONE bf16 part (FGCN_MATH_BF16; reference agcn.py:104-106 under the mixed-precision
    step, session/procedures/step.py:55-78): emb, dS, the on-chip embedding gradient, x and the weights are each rounded to bfloat16
    once, products accumulate in float32 -- against the float64 formulas on operands rounded the same way."""
    from fusion_gcn_amd import ops
    assert ops.emb_tile_available(V, ic, cx)
    emb, ds = rnd(B, T, V, 6 * ic, seed=350), rnd(B, 3, V, V, seed=351, scale=0.3)
    x, base = rnd(B, T, V, cx, seed=352), rnd(B, T, V, cx, seed=353)
    w = rnd(6 * ic, cx, seed=354, scale=(6 * ic) ** -0.5)
    e = bf(emb).reshape(B, T, V, 3, 2, ic)
    theta, phi = e[..., 0, :], e[..., 1, :]
    dtheta = torch.einsum("bkvw,btwke->btvke", bf(ds), phi)
    dphi = torch.einsum("bkvw,btvke->btwke", bf(ds), theta)
    demb = torch.stack([dtheta, dphi], dim=4).reshape(B, T, V, 6 * ic)
    # the kernel rounds the float32 mixing result; the float64 one rounds differently in a few last bits: 2^-9 relative per element
    want_dx = base + bf(demb) @ bf(w)
    want_w = torch.einsum("btvj,btvc->jc", bf(demb), bf(x))
    w3 = ops.pack_split3(gpu(w.reshape(1, 6 * ic, cx)))
    dx = gpu(base)
    ops.emb_dx_tile(gpu(emb), gpu(ds), w3, dx, ic=ic, accumulate=True)
    assert rel_l2(dx.cpu().numpy(), want_dx.numpy()) < 2e-4
    gw, gb = ops.emb_wgrad_tile(gpu(emb), gpu(x), gpu(ds), ic=ic)
    assert rel_l2(gw.cpu().numpy(), want_w.numpy()) < 2e-4
    assert rel_l2(gb.cpu().numpy(), demb.sum((0, 1, 2)).numpy()) < TOL       # the bias gradient sums the float32 mixing result itself
    dx2 = gpu(base)
    ops.emb_dx_tile(gpu(emb), gpu(ds), w3, dx2, ic=ic, accumulate=True)
    assert torch.equal(dx, dx2)


TILE_SHAPES = [(25, 13, 64, 64, 2), (25, 7, 128, 256, 2), (27, 9, 64, 128, 1), (18, 10, 64, 64, 2), (22, 31, 256, 256, 1)]


@pytest.mark.parametrize("V,T,cin,cout,B", TILE_SHAPES)
def test_spatial_tile_kernels_with_one_bf16_part(V, T, cin, cout, B):
    """The three tile kernels of the spatial stage (fgcn_spatial_fwd_tile, fgcn_spatial_bwd_tile, fgcn_spatial_wgrad_tile; reference
    agcn.py:103-111 and its backward) instantiated with ONE bf16 part (FGCN_MATH_BF16, BASELINE config 5): every operand -- x, A^, the
    on-chip aggregation / dagg, dy, the weights -- is rounded to bfloat16 once, products accumulate in float32.  Against the float64
    formulas on operands rounded the same way (the kernel rounds a float32 intermediate, the formula a float64 one: a few last-bit
    differences, 2e-4 in relative L2); bitwise reproducible."""
    from fusion_gcn_amd import ops
    assert ops.spatial_fwd_tile_available(V, cin, cout) and ops.spatial_bwd_tile_available(V, cin, cout) and ops.spatial_wgrad_tile_available(V, cin, cout)
    x, a = rnd(B, T, V, cin, seed=400), rnd(B, 3, V, V, seed=401, scale=0.3)
    wd, bias = rnd(3, cin, cout, seed=402, scale=(3 * cin) ** -0.5), rnd(cout, seed=403)
    dy, base = rnd(B, T, V, cout, seed=404), rnd(B, T, V, cin, seed=405)
    agg = torch.einsum("btvc,bkvw->btwkc", bf(x), bf(a))
    # forward
    want_y = torch.einsum("btwkc,kco->btwo", bf(agg), bf(wd)) + bias
    w3 = ops.pack_split3(gpu(wd.reshape(1, 3 * cin, cout)))
    y, part = ops.spatial_fwd_tile(gpu(x), gpu(a), w3, gpu(bias), Cin=cin, Cout=cout, stats=True)
    assert rel_l2(y.cpu().numpy(), want_y.numpy()) < 2e-4
    assert rel_l2(part.double().sum(0)[0].cpu().numpy(), y.double().sum((0, 1, 2)).cpu().numpy()) < TOL
    y2, _ = ops.spatial_fwd_tile(gpu(x), gpu(a), w3, gpu(bias), Cin=cin, Cout=cout, stats=True)
    assert torch.equal(y, y2)
    # conv_d's weight gradient
    want_w = torch.einsum("btwkc,btwo->kco", bf(agg), bf(dy)).reshape(1, 3 * cin, cout)
    gw = ops.spatial_wgrad_tile(gpu(x), gpu(dy), gpu(a))
    assert rel_l2(gw.cpu().numpy(), want_w.numpy()) < 2e-4
    assert torch.equal(gw, ops.spatial_wgrad_tile(gpu(x), gpu(dy), gpu(a)))
    # backward: dx and the partial grams
    wt = wd.permute(2, 0, 1).reshape(1, cout, 3 * cin)                       # [o][k cin + c]
    dagg = torch.einsum("btwo,okc->btwkc", bf(dy), bf(wt).reshape(cout, 3, cin))
    want_dx = base + torch.einsum("btwkc,bkvw->btvc", bf(dagg), bf(a))
    want_g = torch.einsum("btvc,btwkc->bkvw", bf(x), bf(dagg))
    w3t = ops.pack_split3(gpu(wt))
    dx = gpu(base)
    part_g = ops.spatial_bwd_tile(gpu(dy), gpu(x), gpu(a), w3t, dx, accumulate=True)
    assert rel_l2(dx.cpu().numpy(), want_dx.numpy()) < 2e-4
    assert rel_l2(part_g.double().sum(1)[:, :, :V, :V].cpu().numpy(), want_g.numpy()) < 2e-4
    dx2 = gpu(base)
    part_g2 = ops.spatial_bwd_tile(gpu(dy), gpu(x), gpu(a), w3t, dx2, accumulate=True)
    assert torch.equal(dx, dx2) and torch.equal(part_g, part_g2)


@pytest.mark.parametrize("V,T,cin,ic,B", [(25, 13, 64, 16, 2), (25, 7, 256, 64, 2), (18, 10, 128, 32, 2)])
def test_embedding_forward_tile_form_bf16(V, T, cin, ic, B):
    """fgcn_emb_fwd_tile with ONE bf16 part (FGCN_MATH_BF16; agcn.py:104-106 under the mixed-precision step): x and the weights rounded to
    bfloat16 once, the embedding tile rounded once more for the gram; float32 accumulation."""
    from fusion_gcn_amd import ops
    assert ops.emb_fwd_tile_available(V, ic, cin)
    x, w, bias = rnd(B, T, V, cin, seed=360), rnd(cin, 6 * ic, seed=361, scale=cin ** -0.5), rnd(6 * ic, seed=362)
    want = bf(x) @ bf(w) + bias
    e6 = bf(want).reshape(B, T, V, 3, 2, ic)
    want_s = torch.einsum("btvke,btwke->bkvw", e6[..., 0, :], e6[..., 1, :])
    emb, part = ops.emb_fwd_tile(gpu(x), ops.pack_split3(gpu(w.reshape(1, cin, 6 * ic))), gpu(bias), ic=ic)
    assert rel_l2(emb.cpu().numpy(), want.numpy()) < TOL
    assert rel_l2(part.double().sum(1)[:, :, :V, :V].cpu().numpy(), want_s.numpy()) < 2e-4


# ---- half-precision STORAGE of the temporal conv's operands (include/fgcn.h, the `_h` entry points; paths.half_storage) -------------
@pytest.mark.parametrize("rows,C,res", [(1000, 64, 0), (777, 128, 1), (2048, 256, 2), (50, 8, 1)])
def test_bn_act_and_its_backward_write_bfloat16(rows, C, res):
    """fgcn_bn_act_h / fgcn_bn_act_bwd_apply_h: the bfloat16 tensor they write is the round-to-nearest-even of what the f32 entry
    points write, bit for bit; the sign image and the sums are those of the f32 values."""
    from fusion_gcn_amd import ops
    a, b = gpu(rnd(rows, C, seed=1)), gpu(rnd(rows, C, seed=2))
    mk = lambda seed: gpu(torch.stack([rnd(C, seed=seed), rnd(C, seed=seed + 1).abs() + 0.5, rnd(C, seed=seed + 2), rnd(C, seed=seed + 3)]))  # noqa: E731
    va, vb = mk(10), mk(20)
    args = (a, va, None if res == 0 else b, vb if res == 2 else None)
    want, m0 = ops.bn_act(*args, relu=True, sign_mask=True)
    got, m1 = ops.bn_act(*args, relu=True, sign_mask=True, out_bf16=True)
    assert got.dtype == torch.bfloat16 and torch.equal(got, want.to(torch.bfloat16))
    assert (m0 is None and m1 is None) or torch.equal(m0, m1)
    dout = gpu(rnd(rows, C, seed=3))
    kw = dict(res_mode=res, sign_mask=m0, need_db=res != 0)
    da0, db0, s0 = ops.bn_act_bwd(dout, want, a, va, args[2], args[3], **kw)
    da1, db1, s1 = ops.bn_act_bwd(dout, want, a, va, args[2], args[3], da_bf16=True, **kw)
    assert da1.dtype == torch.bfloat16 and torch.equal(da1, da0.to(torch.bfloat16)) and torch.equal(s0, s1)
    assert (db0 is None and db1 is None) or torch.equal(db0, db1)


@pytest.mark.parametrize("B,T,V,C,N,kt,s", [(2, 20, 25, 64, 64, 9, 1), (2, 21, 25, 128, 128, 9, 2), (3, 13, 18, 256, 256, 9, 1),
                                            (2, 30, 27, 64, 128, 5, 1), (1, 9, 32, 128, 64, 3, 1), (2, 12, 22, 64, 64, 9, 2)])
def test_halo_conv_and_weight_gradient_from_bfloat16_tensors(B, T, V, C, N, kt, s):
    """fgcn_tconv_halo_h (forward, data gradient; strided passes) and fgcn_tconv_wgrad_h on bfloat16 tensors against the f32 entry
    points on the f32 tensors holding the same (bfloat16-representable) values: bit for bit -- the kernels stage the same bytes."""
    from fusion_gcn_amd import block, ops
    from fusion_gcn_amd.packing import Form, Seg
    pad = (kt - 1) // 2
    Tp = (T - 1) // s + 1
    g16 = gpu(rnd(B, T, V, C, seed=5)).to(torch.bfloat16)
    du16 = gpu(rnd(B, Tp, V, N, seed=6)).to(torch.bfloat16)
    g32, du32 = g16.float(), du16.float()
    wt = gpu(rnd(N, C, kt, 1, seed=7, scale=(kt * C) ** -0.5))
    # the block's own packed forms of the temporal weight (forward + data gradient), built as block.pack_weights does
    t_seg = lambda **kw: [Seg(wt, st_tap=1, st_k=kt, st_n=C * kt, klen=C, nlen=N, **kw)]          # noqa: E731  (kt, c, o)
    tt_seg = lambda **kw: [Seg(wt, st_tap=1, st_k=C * kt, st_n=kt, klen=N, nlen=C, **kw)]         # noqa: E731  (kt, o, c)
    from fusion_gcn_amd.packing import PackedWeights
    F = {}
    if s == 1:
        F["t4"] = Form("split3", kt, C, N, t_seg(tlen=kt))
        F["t_t4"] = Form("split3", kt, N, C, tt_seg(tlen=kt))
    else:
        for par, tag in ((0, "e"), (1, "o")):
            n_par = (kt - par + 1) // 2
            F[f"t4_{tag}"] = Form("split3", n_par, C, N, t_seg(tlen=n_par, tap0=par, tap_step=2))
            F[f"t_t4_{tag}"] = Form("split3", n_par, N, C, tt_seg(tlen=n_par, tap0=par, tap_step=2))
    W = PackedWeights(F, wt.device)
    bias = gpu(rnd(N, seed=8))
    outs = []
    for g, du in ((g32, du32), (g16, du16)):
        u = torch.empty(B, Tp, V, N, device=dev())
        part = block.temporal_fwd(g, u, W, bias, kt, s, stats=True)
        dg = torch.zeros(B, T, V, C, device=dev())
        block.temporal_dgrad(du, dg, W, kt, s)
        gw = ops.tconv_wgrad(g, du, taps=kt, stride=s, conv_param=(1, C))
        outs.append((u, part, dg, gw))
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)
    # ... and the values are the convolution of the rounded operands (the f32 call is pinned to float64 by the tests above)
    want = conv_ref(g32.double().cpu(), bf(wt[..., 0].permute(2, 1, 0).cpu()), kt, s, Tp, bias.double().cpu())
    assert rel_l2(outs[1][0].cpu().numpy(), want.numpy()) < TOL


def test_model_step_is_bit_identical_with_half_precision_conv_operands():
    """The whole model in math mode bf16 with G, dU, dY and emb stored as bfloat16 (paths.half_storage) against f32 storage: logits, loss and
    every gradient bit for bit -- the layout change moves bytes, not values.  All ten blocks: strided ones, down / residual convs.
    (paths.half_activations, the default of the mode, goes further and changes values: switched off on both sides here; its own test below.)"""
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from fusion_gcn_amd.loss import cross_entropy
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.util import Graph
    torch.manual_seed(11)
    model = Model((2, 40, 25, 3), 60, Graph(ntu.skeleton_edges, center_joint=ntu.center_joint)).to(dev()).train()
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, "gcn1"):
                m.gcn1.bn.weight.fill_(1.0)
    x = torch.randn(3, 2, 40, 25, 3, device=dev())
    y = torch.randint(0, 60, (3,), device=dev())
    sd = {k: v.clone() for k, v in model.state_dict().items()}

    def run(half):
        model.load_state_dict(sd)
        with ops.context("bf16") as c:
            c.paths.half_storage["bf16"] = half
            c.paths.half_activations["bf16"] = False
            for p in model.parameters():
                p.grad = None
            logits = model(x)
            loss = cross_entropy(logits, y)
            loss.backward()
            return logits.detach().clone(), loss.detach().clone(), [p.grad.clone() for p in model.parameters()]
    l0, s0, g0 = run(False)
    l1, s1, g1 = run(True)
    assert ops.paths().half_storage["bf16"] is True              # the default of the mode
    assert torch.equal(l0, l1) and torch.equal(s0, s1)
    for a_, b_ in zip(g0, g1):
        assert torch.equal(a_, b_)


def test_model_step_with_half_precision_activations():
    """paths.half_activations (the default of math mode bf16): every activation-sized tensor of the training step is a bfloat16 tensor, as in
    the reference's autocast step (session/procedures/step.py:55-78).  (1) The typed kernels really run: the blocks hand bfloat16 tensors
    to each other, Y / U / dG / dx are bfloat16 where the path has the form.  (2) Against the same model with float32 activations: inside
    SURVEY.md section 7's bf16 contract (this random-init model at T = 40, measured on MI355X: logits 2.8e-3, loss 3.7e-3, gradient cosine
    0.992; on the reference's fixture, test_config5_model_against_the_reference: logits 2.2e-3, cosine 0.997 against the float64 oracle) --
    the storage roundings are of the size of the operand roundings the mode already makes.  (3) Two runs agree bit for bit (fixed-order sums)."""
    from fusion_gcn_amd import block, ops
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from fusion_gcn_amd.loss import cross_entropy
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.util import Graph
    torch.manual_seed(11)
    model = Model((2, 40, 25, 3), 60, Graph(ntu.skeleton_edges, center_joint=ntu.center_joint)).to(dev()).train()
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, "gcn1"):
                m.gcn1.bn.weight.fill_(1.0)
    x = torch.randn(3, 2, 40, 25, 3, device=dev())
    y = torch.randint(0, 60, (3,), device=dev())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    seen = {"x16": 0, "dx16": 0, "y16": 0, "u16": 0, "dg16": 0}
    fwd, bwd, conv = block.block_forward, block._block_backward, ops.tconv_halo

    def spy_fwd(x_, *a, **k):
        o, S = fwd(x_, *a, **k)
        seen["x16"] += x_.dtype == torch.bfloat16
        seen["y16"] += S["y"] is not None and S["y"].dtype == torch.bfloat16
        seen["u16"] += S["u"] is not None and S["u"].dtype == torch.bfloat16
        return o, S

    def spy_bwd(d_o, S, *a, **k):
        dx, G = bwd(d_o, S, *a, **k)
        seen["dx16"] += dx is not None and dx.dtype == torch.bfloat16
        return dx, G

    def spy_conv(inp, w4, out, **k):
        seen["dg16"] += k.get("tb") == -1 and out.dtype == torch.bfloat16
        return conv(inp, w4, out, **k)

    def run(half):
        model.load_state_dict(sd)
        with ops.context("bf16") as c:
            c.paths.half_activations["bf16"] = half
            for p in model.parameters():
                p.grad = None
            logits = model(x)
            loss = cross_entropy(logits, y)
            loss.backward()
            return logits.detach().clone(), float(loss), torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
    l0, s0, g0 = run(False)
    block.block_forward, block._block_backward, ops.tconv_halo = spy_fwd, spy_bwd, spy_conv
    try:
        l1, s1, g1 = run(True)
    finally:
        block.block_forward, block._block_backward, ops.tconv_halo = fwd, bwd, conv
    l2, s2, g2 = run(True)
    # ten blocks: nine receive a bfloat16 x (all but the first) and return a bfloat16 dx; Y is bfloat16 in the nine blocks on the tile kernel,
    # U in the eight stride-1 blocks; dG in all ten (eight stride-1 calls + two per strided block)
    assert seen == {"x16": 9, "dx16": 9, "y16": 9, "u16": 8, "dg16": 12}, seen
    e = float((l1 - l0).norm() / l0.norm())
    cos = float(torch.dot(g1, g0) / (g1.norm() * g0.norm()))
    print(f"half-precision activations vs float32 activations (math mode bf16): logits {e:.2e}, |loss diff| {abs(s1 - s0):.2e}, gradient cosine {cos:.4f}")
    assert 0 < e < 1e-2 and abs(s1 - s0) < 1e-2 and cos > 0.98, (e, s1 - s0, cos)
    assert torch.equal(l1, l2) and s1 == s2 and torch.equal(g1, g2)


@pytest.mark.parametrize("V,T,cin,cout,B", [(25, 13, 64, 64, 2), (25, 7, 128, 256, 2), (27, 9, 64, 128, 1), (18, 10, 256, 256, 2), (32, 5, 128, 64, 1)])
def test_spatial_backward_tile_kernels_from_a_bfloat16_dy(V, T, cin, cout, B):
    """fgcn_spatial_bwd_tile_h / fgcn_spatial_wgrad_tile_h on a bfloat16 dy against the f32 entry points on the f32 tensor holding the
    same values: bit for bit (plain, accumulating, gated and per-group-gated forms of the fused backward)."""
    from fusion_gcn_amd import ops
    x, a = gpu(rnd(B, T, V, cin, seed=41)), gpu(rnd(B, 3, V, V, seed=42, scale=0.3))
    dy16 = gpu(rnd(B, T, V, cout, seed=43)).to(torch.bfloat16)
    dy32 = dy16.float()
    w3 = ops.pack_split3(gpu(rnd(1, cout, 3 * cin, seed=44, scale=cout ** -0.5)))
    base = gpu(rnd(B, T, V, cin, seed=45))
    for acc in (False, True):
        d0, d1 = base.clone(), base.clone()
        p0 = ops.spatial_bwd_tile(dy32, x, a, w3, d0, accumulate=acc)
        p1 = ops.spatial_bwd_tile(dy16, x, a, w3, d1, accumulate=acc)
        assert torch.equal(d0, d1) and torch.equal(p0, p1), acc
    if (B * T * V * cin) % 8 == 0:
        e1, e2 = gpu(rnd(B, T, V, cin, seed=46)), gpu(rnd(B, T, V, cin, seed=47))
        m1 = torch.randint(0, 256, (B * T * V * cin // 8,), device=dev(), dtype=torch.uint8)
        m2 = torch.randint(0, 256, (B * T * V * cin // 8,), device=dev(), dtype=torch.uint8)
        eg = gpu(rnd(B, cin, seed=48))
        for gated in ([(e1, m1), (e2, m2)], [(eg, m1, 1), (e2, m2)]):
            d0, d1 = torch.empty_like(base), torch.empty_like(base)
            p0 = ops.spatial_bwd_tile(dy32, x, a, w3, d0, accumulate=False, gated=gated)
            p1 = ops.spatial_bwd_tile(dy16, x, a, w3, d1, accumulate=False, gated=gated)
            assert torch.equal(d0, d1) and torch.equal(p0, p1)
    assert torch.equal(ops.spatial_wgrad_tile(x, dy32, a), ops.spatial_wgrad_tile(x, dy16, a))
    assert torch.equal(ops.spatial_wgrad_tile(x, dy32, a[:1]), ops.spatial_wgrad_tile(x, dy16, a[:1]))


def test_inference_kernels_in_bf16_mode():
    """The inference forms of the two north-star kernels (BatchNorm + shortcut + ReLU in the epilogue) with ONE bf16 part: the model's
    eval logits under torch.no_grad() against the two-pass eval path of the same mode (bf16-level agreement: the epilogue's f32
    arithmetic differs in its last bits and every later staging rounds to bfloat16 again)."""
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.util import Graph
    torch.manual_seed(13)
    model = Model((2, 40, 25, 3), 60, Graph(ntu.skeleton_edges, center_joint=ntu.center_joint)).to(dev())
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, "gcn1"):
                m.gcn1.bn.weight.fill_(1.0)
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    model.eval()
    x = torch.randn(3, 2, 40, 25, 3, device=dev())
    seen = []
    conv = ops.tconv_halo_bn_relu
    ops.tconv_halo_bn_relu = lambda *a, **k: (seen.append(1), conv(*a, **k))[1]
    try:
        with torch.no_grad():
            with ops.context() as c:
                c.paths.fused_inference = False
                ref = model(x)
            assert not seen
            fused = model(x)
        assert len(seen) == 7
    finally:
        ops.tconv_halo_bn_relu = conv
    err = float((fused - ref).norm() / ref.norm())
    assert err < 5e-2, err      # (two realisations of ten blocks' bf16 roundings: see test_config5_model_against_the_reference)


@pytest.mark.parametrize("V,T,cin,ic,B", [(25, 13, 64, 16, 2), (25, 7, 128, 32, 2), (27, 9, 64, 32, 1), (18, 10, 128, 64, 2), (22, 31, 256, 64, 1)])
def test_embedding_kernels_with_a_bfloat16_emb(V, T, cin, ic, B):
    """fgcn_emb_fwd_tile_h writes the bfloat16 rounding of what fgcn_emb_fwd_tile writes (same gram partials); fgcn_emb_dx_tile_h /
    fgcn_emb_wgrad_tile_h on that tensor equal the f32 entry points on the f32 tensor holding the same values, bit for bit."""
    from fusion_gcn_amd import ops
    x = gpu(rnd(B, T, V, cin, seed=61))
    w3 = ops.pack_split3(gpu(rnd(1, cin, 6 * ic, seed=62, scale=cin ** -0.5)))
    bias = gpu(rnd(6 * ic, seed=63))
    e32, p32 = ops.emb_fwd_tile(x, w3, bias, ic=ic)
    e16, p16 = ops.emb_fwd_tile(x, w3, bias, ic=ic, emb_bf16=True)
    assert e16.dtype == torch.bfloat16 and torch.equal(e16, e32.to(torch.bfloat16)) and torch.equal(p32, p16)
    ef = e16.float()                                              # the f32 tensor holding the same (bfloat16-representable) values
    ds = gpu(rnd(B, 3, V, V, seed=64, scale=0.2))
    wt = ops.pack_split3(gpu(rnd(1, 6 * ic, cin, seed=65, scale=(6 * ic) ** -0.5)))
    base = gpu(rnd(B, T, V, cin, seed=66))
    for acc in (False, True):
        d0, d1 = base.clone(), base.clone()
        ops.emb_dx_tile(ef, ds, wt, d0, ic=ic, accumulate=acc)
        ops.emb_dx_tile(e16, ds, wt, d1, ic=ic, accumulate=acc)
        assert torch.equal(d0, d1), acc
    gw0, gb0 = ops.emb_wgrad_tile(ef, x, ds, ic=ic)
    gw1, gb1 = ops.emb_wgrad_tile(e16, x, ds, ic=ic)
    assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1)
    gw2, gb2 = ops.emb_wgrad_tile(e16, x, ds[:1], ic=ic)          # shared dS
    gw3, gb3 = ops.emb_wgrad_tile(ef, x, ds[:1], ic=ic)
    assert torch.equal(gw2, gw3) and torch.equal(gb2, gb3)


# ---- half-precision ACTIVATION storage (include/fgcn.h, the typed `_t` entry points; paths.half_activations) ---------------------------
# A bfloat16 INPUT of a kernel must give the bits of the float32 call on the same values; a bfloat16 OUTPUT must be the round-to-nearest-even
# of the float32 call's output, with BatchNorm sums of the float32 values.
def h16(*shape, seed=0, scale=1.0):
    return gpu(rnd(*shape, seed=seed, scale=scale)).to(torch.bfloat16)


@pytest.mark.parametrize("rows,C,res", [(1000, 64, 0), (777, 128, 1), (2048, 256, 2), (50, 8, 1), (600, 64, 2), (1000, 12, 1), (500, 20, 2)])   # (C % 8 != 0: the four-wide typed kernels)
def test_batchnorm_passes_with_bfloat16_operands(rows, C, res):
    """fgcn_bn_act_t / fgcn_bn_act_pool_t / fgcn_bn_act_bwd_reduce_t / fgcn_bn_act_bwd_apply_t with a, the shortcut and the incoming
    gradient as bfloat16 tensors, in every combination the block produces, against the float32 entry points on the same values."""
    from fusion_gcn_amd import ops
    a16, b16, d16 = h16(rows, C, seed=1), h16(rows, C, seed=2), h16(rows, C, seed=3)
    mk = lambda seed: gpu(torch.stack([rnd(C, seed=seed), rnd(C, seed=seed + 1).abs() + 0.5, rnd(C, seed=seed + 2), rnd(C, seed=seed + 3)]))  # noqa: E731
    va, vb = mk(10), mk(20)
    for a_half, b_half, o_half in ((True, True, True), (True, False, False), (False, True, True), (True, True, False)):
        a = a16 if a_half else a16.float()
        b = None if res == 0 else (b16 if b_half else b16.float())
        vec_b = vb if res == 2 else None
        want, m0 = ops.bn_act(a16.float(), va, None if res == 0 else b16.float(), vec_b, relu=True, sign_mask=True)
        got, m1 = ops.bn_act(a, va, b, vec_b, relu=True, sign_mask=True, out_bf16=o_half)
        assert torch.equal(got, want.to(got.dtype)) and ((m0 is None and m1 is None) or torch.equal(m0, m1)), (a_half, b_half, o_half)
        if m0 is None:
            continue
        for d_half in (True, False):
            dout = d16 if d_half else d16.float()
            kw = dict(res_mode=res, sign_mask=m0, need_db=res == 2)
            da0, db0, s0 = ops.bn_act_bwd(d16.float(), None, a16.float(), va, None if res == 0 else b16.float(), vec_b, **kw)
            da1, db1, s1 = ops.bn_act_bwd(dout, None, a, va, b, vec_b, da_bf16=o_half, **kw)
            assert torch.equal(da1, da0.to(da1.dtype)) and torch.equal(s0, s1), (a_half, b_half, o_half, d_half)
            assert (db0 is None and db1 is None) or torch.equal(db0, db1)
    if C % 8 == 0 and rows % 4 == 0:
        for a_half, b_half in ((True, True), (True, False), (False, True)):
            args32 = (a16.float(), va, None if res == 0 else b16.float(), vb if res == 2 else None)
            args = (a16 if a_half else a16.float(), va, None if res == 0 else (b16 if b_half else b16.float()), vb if res == 2 else None)
            p0, m0 = ops.bn_act_pool(*args32, 4)
            p1, m1 = ops.bn_act_pool(*args, 4)
            assert torch.equal(p0, p1) and torch.equal(m0, m1)


@pytest.mark.parametrize("B,T,V,C,N,kt,s", [(2, 20, 25, 64, 64, 9, 1), (3, 13, 18, 256, 256, 9, 1), (2, 30, 27, 64, 128, 5, 1), (1, 9, 32, 128, 64, 3, 1),
                                            (2, 21, 25, 128, 128, 9, 2), (8, 40, 25, 128, 128, 9, 1), (16, 64, 25, 64, 64, 9, 1)])
def test_halo_conv_writes_bfloat16(B, T, V, C, N, kt, s):
    """fgcn_tconv_halo_t with mask 3: the output is the rounding of the float32 output of fgcn_tconv_halo_h, the BatchNorm moments are
    those of the float32 values, bit for bit (forward with statistics; data gradient, also through the frame views of a strided conv)."""
    from fusion_gcn_amd import ops
    pad = (kt - 1) // 2
    g16 = h16(B, T, V, C, seed=5)
    w = ops.pack_split3(gpu(rnd(kt, C, N, seed=7, scale=(kt * C) ** -0.5)))
    bias = gpu(rnd(N, seed=8))
    if s == 1:
        u32 = torch.empty(B, T, V, N, device=dev())
        u16 = torch.empty(B, T, V, N, device=dev(), dtype=torch.bfloat16)
        p32 = ops.tconv_halo(g16, w, u32, Th=T, taps=kt, tb=1, tc=-pad, bias=bias, stats=True)
        p16 = ops.tconv_halo(g16, w, u16, Th=T, taps=kt, tb=1, tc=-pad, bias=bias, stats=True)
        assert torch.equal(u16, u32.to(torch.bfloat16)) and torch.equal(p32, p16)
        ops.tconv_halo(g16, w, u32, Th=T, taps=kt, tb=-1, tc=pad)
        ops.tconv_halo(g16, w, u16, Th=T, taps=kt, tb=-1, tc=pad)
        assert torch.equal(u16, u32.to(torch.bfloat16))
    else:           # the data gradient of a stride-2 conv: even / odd output frames through out_view
        Tp = (T - 1) // 2 + 1
        du16 = h16(B, Tp, V, C, seed=6)
        we = ops.pack_split3(gpu(rnd((kt + 1) // 2, C, N, seed=9, scale=(kt * C) ** -0.5)))
        d32 = torch.zeros(B, T, V, N, device=dev())
        d16 = torch.zeros(B, T, V, N, device=dev(), dtype=torch.bfloat16)
        for out in (d32, d16):
            ops.tconv_halo(du16, we, out, Th=(T + 1) // 2, taps=(kt + 1) // 2, tb=-1, tc=pad // 2, out_view=(2, 0))
        assert torch.equal(d16, d32.to(torch.bfloat16))


@pytest.mark.parametrize("V,T,cin,cout,B", [(25, 13, 64, 64, 2), (25, 7, 128, 256, 2), (27, 9, 64, 128, 1), (18, 10, 256, 256, 2), (32, 5, 128, 64, 1), (22, 40, 64, 64, 3)])
def test_spatial_tile_kernels_on_bfloat16_activations(V, T, cin, cout, B):
    """fgcn_spatial_fwd_tile_t (x in, y out), fgcn_spatial_wgrad_tile_t (x, dy), fgcn_spatial_bwd_tile_t (dy, x, dx, gated addends) against
    the float32 / `_h` forms on the same values: equal inputs give equal bits, a bfloat16 output is the rounded float32 output."""
    from fusion_gcn_amd import ops
    x16, a = h16(B, T, V, cin, seed=41), gpu(rnd(B, 3, V, V, seed=42, scale=0.3))
    x32 = x16.float()
    wd = ops.pack_split3(gpu(rnd(1, 3 * cin, cout, seed=40, scale=(3 * cin) ** -0.5)))
    bias = gpu(rnd(cout, seed=39))
    y32, p32 = ops.spatial_fwd_tile(x32, a, wd, bias, Cin=cin, Cout=cout)
    for xin in (x16, x32):
        y16, p16 = ops.spatial_fwd_tile(xin, a, wd, bias, Cin=cin, Cout=cout, y_bf16=True)
        assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.to(torch.bfloat16)) and torch.equal(p32, p16)
    dy16 = h16(B, T, V, cout, seed=43)
    assert torch.equal(ops.spatial_wgrad_tile(x32, dy16, a), ops.spatial_wgrad_tile(x16, dy16, a))
    assert torch.equal(ops.spatial_wgrad_tile(x32, dy16, a[:1]), ops.spatial_wgrad_tile(x16, dy16, a[:1]))
    w3 = ops.pack_split3(gpu(rnd(1, cout, 3 * cin, seed=44, scale=cout ** -0.5)))
    base16 = h16(B, T, V, cin, seed=45)
    for acc in (False, True):
        d0, d1 = base16.float(), base16.clone()
        p0 = ops.spatial_bwd_tile(dy16, x32, a, w3, d0, accumulate=acc)
        p1 = ops.spatial_bwd_tile(dy16, x16, a, w3, d1, accumulate=acc)
        assert torch.equal(d1, d0.to(torch.bfloat16)) and torch.equal(p0, p1), acc
    if (B * T * V * cin) % 8 == 0:
        e1, e2 = h16(B, T, V, cin, seed=46), h16(B, T, V, cin, seed=47)
        m1 = torch.randint(0, 256, (B * T * V * cin // 8,), device=dev(), dtype=torch.uint8)
        m2 = torch.randint(0, 256, (B * T * V * cin // 8,), device=dev(), dtype=torch.uint8)
        eg = gpu(rnd(B, cin, seed=48))
        for g32, g16 in (([(e1.float(), m1), (e2.float(), m2)], [(e1, m1), (e2, m2)]), ([(eg, m1, 1), (e2.float(), m2)], [(eg, m1, 1), (e2, m2)])):
            d0, d1 = torch.empty_like(x32), torch.empty_like(x16)
            p0 = ops.spatial_bwd_tile(dy16, x32, a, w3, d0, accumulate=False, gated=g32)
            p1 = ops.spatial_bwd_tile(dy16, x16, a, w3, d1, accumulate=False, gated=g16)
            assert torch.equal(d1, d0.to(torch.bfloat16)) and torch.equal(p0, p1)


@pytest.mark.parametrize("V,T,cin,ic,B", [(25, 13, 64, 16, 2), (25, 7, 128, 32, 2), (27, 9, 64, 32, 1), (18, 10, 128, 64, 2), (22, 31, 256, 64, 1)])
def test_embedding_kernels_on_bfloat16_activations(V, T, cin, ic, B):
    """fgcn_emb_fwd_tile_t (x in), fgcn_emb_dx_tile_t (dx read-modify-written as bfloat16), fgcn_emb_wgrad_tile_t (x in) against the `_h`
    forms on the same values."""
    from fusion_gcn_amd import ops
    x16 = h16(B, T, V, cin, seed=61)
    x32 = x16.float()
    w3 = ops.pack_split3(gpu(rnd(1, cin, 6 * ic, seed=62, scale=cin ** -0.5)))
    bias = gpu(rnd(6 * ic, seed=63))
    e0, p0 = ops.emb_fwd_tile(x32, w3, bias, ic=ic, emb_bf16=True)
    e1, p1 = ops.emb_fwd_tile(x16, w3, bias, ic=ic, emb_bf16=True)
    assert e1.dtype == torch.bfloat16 and torch.equal(e0, e1) and torch.equal(p0, p1)
    e2, p2 = ops.emb_fwd_tile(x16, w3, bias, ic=ic)                 # a float32 emb from a bfloat16 x
    e3, _ = ops.emb_fwd_tile(x32, w3, bias, ic=ic)
    assert e2.dtype == torch.float32 and torch.equal(e2, e3) and torch.equal(p2, p0)
    ds = gpu(rnd(B, 3, V, V, seed=64, scale=0.2))
    wt = ops.pack_split3(gpu(rnd(1, 6 * ic, cin, seed=65, scale=(6 * ic) ** -0.5)))
    base16 = h16(B, T, V, cin, seed=66)
    for acc in (False, True):
        d0, d1 = base16.float(), base16.clone()
        ops.emb_dx_tile(e1, ds, wt, d0, ic=ic, accumulate=acc)
        ops.emb_dx_tile(e1, ds, wt, d1, ic=ic, accumulate=acc)
        assert torch.equal(d1, d0.to(torch.bfloat16)), acc
    # dx = bfloat16(dx_old + term): float32 old values, bfloat16 result (the last writer of a float32-accumulated gradient converts on the way)
    old = gpu(rnd(B, T, V, cin, seed=67))
    d0, d1 = old.clone(), torch.empty(B, T, V, cin, device=dev(), dtype=torch.bfloat16)
    ops.emb_dx_tile(e1, ds, wt, d0, ic=ic, accumulate=True)
    ops.emb_dx_tile(e1, ds, wt, d1, ic=ic, accumulate=True, dx_old=old)
    assert torch.equal(d1, d0.to(torch.bfloat16))
    for d in (ds, ds[:1]):
        gw0, gb0 = ops.emb_wgrad_tile(e1, x32, d, ic=ic)
        gw1, gb1 = ops.emb_wgrad_tile(e1, x16, d, ic=ic)
        assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1)


@pytest.mark.parametrize("B,T,V,K,N,s", [(2, 20, 25, 64, 128, 1), (2, 21, 25, 64, 128, 2), (3, 10, 18, 128, 256, 2), (2, 9, 27, 4, 64, 1), (2, 40, 25, 128, 64, 1),
                                         (1, 300, 25, 256, 128, 1)])
def test_row_gemms_on_bfloat16_activations(B, T, V, K, N, s):
    """fgcn_rows_gemm_t / fgcn_pw_gemm_t (the shortcut convolutions of the blocks that change width or stride): a bfloat16 input gives the bits
    of the float32 call on the same values, a bfloat16 output is the rounded float32 output with the float32 BatchNorm sums; also the
    accumulating data-gradient form from a bfloat16 gradient."""
    from fusion_gcn_amd import ops
    Tp = (T - 1) // s + 1
    x16 = h16(B, T, V, K, seed=71)
    x32 = x16.float()
    w = gpu(rnd(1, K, N, seed=72, scale=K ** -0.5))
    bias = gpu(rnd(N, seed=73))
    tmap = (1, s, 0, 0, 1)
    o32 = torch.empty(B, Tp, V, N, device=dev())
    p32 = ops.rows_gemm(x32, w, o32, K=K, N=N, tmap=tmap, bias=bias, stats=True)
    for xin, half_out in ((x16, False), (x16, True), (x32, True)):
        o = torch.empty(B, Tp, V, N, device=dev(), dtype=torch.bfloat16 if half_out else torch.float32)
        p = ops.rows_gemm(xin, w, o, K=K, N=N, tmap=tmap, bias=bias, stats=True)
        assert torch.equal(o, o32.to(o.dtype)) and torch.equal(p, p32), (xin.dtype, half_out)
    # the data gradient of that convolution, accumulated into a float32 dx from a bfloat16 gradient
    g16 = h16(B, Tp, V, N, seed=74)
    wt = gpu(rnd(1, N, K, seed=75, scale=N ** -0.5))
    base = gpu(rnd(B, T, V, K, seed=76))
    d0, d1 = base.clone(), base.clone()
    ops.rows_gemm(g16.float(), wt, d0, K=N, N=K, tmap=(1, 1, 0, 0, s), accumulate=True)
    ops.rows_gemm(g16, wt, d1, K=N, N=K, tmap=(1, 1, 0, 0, s), accumulate=True)
    assert torch.equal(d0, d1)
    if s == 1 and K % 32 == 0:
        w3 = ops.pack_split3(w)
        o32 = torch.empty(B, T, V, N, device=dev())
        p32 = ops.pw_gemm(x32, w3, o32, bias=bias, stats=True)
        for xin, half_out in ((x16, False), (x16, True), (x32, True)):
            o = torch.empty(B, T, V, N, device=dev(), dtype=torch.bfloat16 if half_out else torch.float32)
            p = ops.pw_gemm(xin, w3, o, bias=bias, stats=True)
            # (the sums of squares: the compiler contracts `s += v * v` to an fma in one instantiation and not in the other -- last-bit differences)
            assert torch.equal(o, o32.to(o.dtype)) and torch.allclose(p, p32, rtol=2e-6, atol=0), (xin.dtype, half_out)
        w3t = ops.pack_split3(wt)
        d0, d1 = base.clone(), base.clone()
        ops.pw_gemm(g16.float(), w3t, d0, accumulate=True)
        ops.pw_gemm(g16, w3t, d1, accumulate=True)
        assert torch.equal(d0, d1)


@pytest.mark.parametrize("B,T,V,cin,cout,s", [(2, 20, 25, 64, 128, 1), (2, 21, 25, 64, 128, 2), (2, 12, 18, 128, 256, 2), (1, 30, 27, 128, 64, 1)])
def test_shortcut_branch_backward_on_bfloat16_tensors(B, T, V, cin, cout, s):
    """The backward of the blocks that change width or stride under half-precision activation storage: the shortcut BatchNorm's gradient as a
    bfloat16 tensor (fgcn_bn_act_bwd_apply_t, bit 4), the 1x1 weight gradient from bfloat16 operands (fgcn_pw_wgrad_h), and the fused spatial
    backward reading a bfloat16 x while it accumulates into a float32 dx (fgcn_spatial_bwd_tile_t, mask 3) -- against the float32 forms."""
    from fusion_gcn_amd import ops
    Tp = (T - 1) // s + 1
    rows, C = B * Tp * V, cout
    a16, b16, d16 = h16(rows, C, seed=81), h16(rows, C, seed=82), h16(rows, C, seed=83)
    mk = lambda seed: gpu(torch.stack([rnd(C, seed=seed), rnd(C, seed=seed + 1).abs() + 0.5, rnd(C, seed=seed + 2), rnd(C, seed=seed + 3)]))  # noqa: E731
    va, vb = mk(10), mk(20)
    _, m0 = ops.bn_act(a16.float(), va, b16.float(), vb, relu=True, sign_mask=True)
    kw = dict(res_mode=2, sign_mask=m0)
    da0, db0, s0 = ops.bn_act_bwd(d16.float(), None, a16.float(), va, b16.float(), vb, **kw)
    da1, db1, s1 = ops.bn_act_bwd(d16, None, a16, va, b16, vb, da_bf16=True, db_bf16=True, **kw)
    assert db1.dtype == torch.bfloat16 and torch.equal(db1, db0.to(torch.bfloat16)) and torch.equal(da1, da0.to(torch.bfloat16)) and torch.equal(s0, s1)
    # the 1x1 (strided) weight gradient
    x16, g16 = h16(B, T, V, cin, seed=84), h16(B, Tp, V, cout, seed=85)
    gw0 = ops.rows_wgrad(x16.float(), g16.float(), K=cin, N=cout, tmap=(1, s, 0, 0, 1), conv_param=(1, cin))
    gw1 = ops.rows_wgrad(x16, g16, K=cin, N=cout, tmap=(1, s, 0, 0, 1), conv_param=(1, cin))
    assert torch.equal(gw0, gw1)
    # the fused spatial backward: bfloat16 dy and x, float32 dx (plain and accumulating)
    a = gpu(rnd(B, 3, V, V, seed=86, scale=0.3))
    dy16 = h16(B, T, V, cout, seed=87)
    w3 = ops.pack_split3(gpu(rnd(1, cout, 3 * cin, seed=88, scale=cout ** -0.5)))
    base = gpu(rnd(B, T, V, cin, seed=89))
    for acc in (False, True):
        e0, e1 = base.clone(), base.clone()
        p0 = ops.spatial_bwd_tile(dy16, x16.float(), a, w3, e0, accumulate=acc)
        p1 = ops.spatial_bwd_tile(dy16, x16, a, w3, e1, accumulate=acc)
        assert torch.equal(e0, e1) and torch.equal(p0, p1), acc
