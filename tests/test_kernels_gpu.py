"""Per-kernel parity on the MI355X: each libfgcn entry point (through the ctypes wrappers in fusion_gcn_amd.ops)
against a float64 torch restatement of the same formula on identical seeded inputs.  Forward-type kernels must
agree to 2e-6 relative L2 (f32 MFMA = exact f32 FMA chains), reductions over ~1e5 rows to 1e-5."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

FWD_TOL = 3e-6
RED_TOL = 2e-5


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


def to_gpu(t):
    return t.float().to(dev()).contiguous()


def ref_rows_conv(x, w, tmap, T_out, bias=None):
    """x (B,T_in,V,K) f64, w (taps,K,N): out[b,to,v,:] = sum_j x[b,ti(to,j),v,:] @ w[j]."""
    taps, ta, tb, tc, td = tmap
    B, T_in, V, K = x.shape
    out = torch.zeros(B, T_out, V, w.shape[2], dtype=torch.float64)
    for to in range(T_out):
        for j in range(taps):
            num = to * ta + j * tb + tc
            if num < 0 or num % td:
                continue
            ti = num // td
            if ti >= T_in:
                continue
            out[:, to] += x[:, ti] @ w[j]
    if bias is not None:
        out += bias
    return out


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,V,K,N,kt,s", [
    (2, 20, 25, 64, 64, 9, 1),      # temporal conv, stride 1
    (2, 21, 25, 64, 128, 9, 2),     # stride 2, odd T
    (3, 10, 18, 128, 96, 1, 1),     # theta|phi embedding shape (N = 96)
    (2, 12, 22, 256, 384, 1, 1),    # 6ic = 384
    (2, 9, 27, 4, 64, 1, 1),        # network input (3 channels padded to 4)
    (1, 30, 25, 192, 64, 1, 1),     # conv_d on stacked agg (K = 3*64)
    (2, 16, 25, 64, 4, 1, 1),       # data gradient into the padded 4-channel input
    (1, 7, 5, 40, 36, 3, 1),        # ragged K / N (not multiples of 32), 3 taps
])
def test_rows_gemm_forward_and_stats(B, T, V, K, N, kt, s):
    from fusion_gcn_amd import ops
    tmap = ops.conv_tmap(kt, s)
    T_out = (T - 1) // s + 1
    x, w, b = rnd(B, T, V, K, seed=1), rnd(kt, K, N, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    want = ref_rows_conv(x, w, tmap, T_out, b)
    out = torch.empty(B, T_out, V, N, device=dev())
    part = ops.rows_gemm(to_gpu(x), to_gpu(w), out, K=K, N=N, tmap=tmap, bias=to_gpu(b), stats=True)
    assert rel_l2(out.cpu().numpy(), want.numpy()) < FWD_TOL
    tot = part.double().sum(0).cpu()
    flat = want.reshape(-1, N)
    assert rel_l2(tot[0].numpy(), flat.sum(0).numpy()) < RED_TOL
    assert rel_l2(tot[1].numpy(), (flat ** 2).sum(0).numpy()) < RED_TOL


def test_rows_gemm_channel_windows_and_accumulate():
    from fusion_gcn_amd import ops
    B, T, V = 2, 6, 25
    x = rnd(B, T, V, 96, seed=4)
    w = rnd(1, 32, 64, seed=5, scale=0.2)
    base = rnd(B, T, V, 128, seed=6)
    out = to_gpu(base)
    ops.rows_gemm(to_gpu(x), to_gpu(w), out, K=32, N=64, in_coff=64, out_coff=32, accumulate=True)
    want = base.clone()
    want[..., 32:96] += x[..., 64:96] @ w[0]
    assert rel_l2(out.cpu().numpy(), want.numpy()) < FWD_TOL


@pytest.mark.parametrize("kt,s,T", [(9, 1, 20), (9, 2, 21), (9, 2, 20), (1, 2, 11)])
def test_rows_gemm_data_gradient_map(kt, s, T):
    """conv_dgrad_tmap reproduces autograd's input gradient of the strided temporal conv."""
    from fusion_gcn_amd import ops
    B, V, C, O = 2, 5, 32, 64
    T_out = (T - 1) // s + 1
    w = rnd(kt, C, O, seed=7, scale=0.1)                       # forward packed (kt, c, o)
    x = rnd(B, T, V, C, seed=8).requires_grad_(True)
    y = ref_rows_conv(x, w, ops.conv_tmap(kt, s), T_out)
    dy = rnd(B, T_out, V, O, seed=9)
    (dx_want,) = torch.autograd.grad((y * dy).sum(), x)
    dx = torch.empty(B, T, V, C, device=dev())
    ops.rows_gemm(to_gpu(dy), to_gpu(w.permute(0, 2, 1)), dx, K=O, N=C, tmap=ops.conv_dgrad_tmap(kt, s))
    assert rel_l2(dx.cpu().numpy(), dx_want.numpy()) < FWD_TOL


@pytest.mark.parametrize("B,T,V,K,N,kt,s", [
    (3, 20, 25, 64, 64, 9, 1),      # two row halves per stage (N <= 64), 9 taps
    (2, 21, 25, 64, 128, 9, 2),     # stride 2, odd T: even frames 5 taps + odd frames 4 taps
    (2, 20, 27, 128, 128, 9, 2),    # stride 2, even T, V = 27 (largest window)
    (2, 13, 18, 128, 256, 9, 1),    # two 128-column tiles
    (1, 2, 25, 32, 96, 9, 1),       # a sample shorter than one stage, N tail inside a 128 tile
    (2, 9, 20, 40, 36, 3, 1),       # ragged K / N, 3 taps
    (2, 7, 32, 64, 64, 5, 1),       # V = 32, 5 taps
    (3, 40, 25, 256, 256, 9, 1),    # 256 channels
    (2, 11, 25, 64, 64, 7, 1),      # 7 taps: not instantiated -> falls back to the per-tap kernel
])
def test_tconv_wgrad_all_taps_in_one_pass(B, T, V, K, N, kt, s):
    """The multi-tap temporal weight gradient equals autograd's conv weight gradient (and the per-tap kernel)."""
    from fusion_gcn_amd import ops
    T_out = (T - 1) // s + 1
    x = rnd(B, T, V, K, seed=31)
    w = rnd(kt, K, N, seed=32, scale=0.1).requires_grad_(True)
    dy = rnd(B, T_out, V, N, seed=33)
    y = ref_rows_conv(x, w, ops.conv_tmap(kt, s), T_out)
    (dw_want,) = torch.autograd.grad((y * dy).sum(), w)
    got = ops.tconv_wgrad(to_gpu(x), to_gpu(dy), taps=kt, stride=s, all_taps=True)
    assert rel_l2(got.cpu().numpy(), dw_want.numpy()) < RED_TOL
    ref = ops.rows_wgrad(to_gpu(x), to_gpu(dy), K=K, N=N, tmap=ops.conv_tmap(kt, s))
    assert rel_l2(got.cpu().numpy(), ref.cpu().numpy()) < RED_TOL
    again = ops.tconv_wgrad(to_gpu(x), to_gpu(dy), taps=kt, stride=s, all_taps=True)
    assert torch.equal(got, again)          # fixed-order slabs: bitwise reproducible
    acc = got.clone()
    ops.tconv_wgrad(to_gpu(x), to_gpu(dy), taps=kt, stride=s, out=acc, accumulate=True, all_taps=True)
    assert rel_l2(acc.cpu().numpy(), 2 * dw_want.numpy()) < RED_TOL


def test_all_taps_weight_gradient_over_random_shapes():
    """The all-taps weight gradient over 24 seeded random shapes (joint counts 5..32, 3 / 5 / 9 taps, stride 1 / 2, sample lengths
    from below one stage of rows to several, tile-filling and ragged channel counts) against float64 sums: the split-bf16 kernel
    keeps its tap window in a circular LDS image that is filled at the first stage of a sample or of a workgroup's share and
    extended by the new rows of every later stage."""
    import random
    from fusion_gcn_amd import ops
    rng = random.Random(3)
    gen = torch.Generator().manual_seed(3)
    for _ in range(24):
        V = rng.choice([5, 18, 20, 22, 25, 27, 32])
        B, kt, s = rng.randint(1, 5), rng.choice([9, 9, 5, 3]), rng.choice([1, 1, 2])
        T = rng.randint(max(2, kt // 2 + 1), 40)
        K, N = rng.choice([32, 64, 128, 160, 256]), rng.choice([32, 64, 96, 128, 256])
        Tg, pad = (T - 1) // s + 1, (kt - 1) // 2
        a, g = torch.randn(B, T, V, K, generator=gen), torch.randn(B, Tg, V, N, generator=gen)
        ap = torch.zeros(B, T + 2 * pad, V, K, dtype=torch.float64)
        ap[:, pad:pad + T] = a.double()
        want = torch.stack([torch.einsum("btvk,btvn->kn", ap[:, j:j + (Tg - 1) * s + 1:s], g.double()) for j in range(kt)])
        got = ops.tconv_wgrad(to_gpu(a), to_gpu(g), taps=kt, stride=s, all_taps=True)
        assert rel_l2(got.cpu().numpy(), want.numpy()) < RED_TOL, (B, T, V, K, N, kt, s)


@pytest.mark.parametrize("B,T,V,C,O,s", [(3, 20, 25, 64, 64, 1), (2, 21, 25, 64, 128, 2), (2, 20, 27, 128, 128, 2),
                                         (2, 13, 18, 128, 256, 1), (3, 7, 32, 32, 96, 1), (2, 1, 25, 64, 64, 2),
                                         (1, 40, 25, 256, 256, 1), (2, 9, 20, 64, 64, 2)])
def test_halo_temporal_conv_forward_and_data_gradient(B, T, V, C, O, s):
    """The halo-tile kernel (input staged once per channel chunk, parity-split strides) vs the per-frame formula,
    through the same helpers the block uses; tiles straddle sample boundaries (T*V is not a multiple of 128)."""
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.block import temporal_dgrad, temporal_fwd
    kt = 9
    Tp = (T - 1) // s + 1
    wt = rnd(kt, C, O, seed=80, scale=(kt * C) ** -0.5)           # (kt, c, o)
    bias = rnd(O, seed=81)
    wg, wg_t = to_gpu(wt), to_gpu(wt.permute(0, 2, 1))
    W = {"t": wg, "t_t": wg_t}
    if s == 1:
        W["t4"], W["t_t4"] = ops.pack_conv(wg), ops.pack_conv(wg_t)        # the streamed form of the current math mode
    else:
        for par, tag in ((0, "e"), (1, "o")):
            W[f"t4_{tag}"] = ops.pack_conv(wg[par::2].contiguous())
            W[f"t_t4_{tag}"] = ops.pack_conv(wg_t[par::2].contiguous())
    x = rnd(B, T, V, C, seed=82).requires_grad_(True)
    want = ref_rows_conv(x, wt, ops.conv_tmap(kt, s), Tp, bias)
    u = torch.full((B, Tp, V, O), 3.0, device=dev())
    part = temporal_fwd(to_gpu(x.detach()), u, W, to_gpu(bias), kt, s, stats=True)
    assert rel_l2(u.cpu().numpy(), want.detach().numpy()) < FWD_TOL
    tot = part.double().sum(0).cpu()
    flat = want.detach().reshape(-1, O)
    assert rel_l2(tot[0].numpy(), flat.sum(0).numpy()) < RED_TOL
    assert rel_l2(tot[1].numpy(), (flat ** 2).sum(0).numpy()) < RED_TOL
    du = rnd(B, Tp, V, O, seed=83)
    (dx_want,) = torch.autograd.grad((want * du).sum(), x)
    dg = torch.full((B, T, V, C), 5.0, device=dev())
    temporal_dgrad(to_gpu(du), dg, W, kt, s)
    assert rel_l2(dg.cpu().numpy(), dx_want.numpy()) < FWD_TOL


@pytest.mark.math_modes("bf16x3", "f16x2")
@pytest.mark.parametrize("rows,K,N,ld_in,ld_out", [
    (3000, 64, 96, 64, 96), (2999, 96, 64, 96, 64), (1283, 128, 384, 128, 384), (517, 192, 128, 192, 128), (4097, 256, 768, 256, 768),
    (130, 384, 256, 384, 256), (1000, 64, 128, 64, 128), (127, 32, 4, 36, 8), (12000, 64, 192, 64, 192), (70000, 64, 64, 64, 64),
    (700, 96, 192, 96, 192), (300, 32, 128, 32, 128), (513, 160, 132, 160, 136)])
def test_persistent_pointwise_gemm(rows, K, N, ld_in, ld_out):
    """ops.pw_gemm (fgcn_pw.hip): the block's 1x1 convolutions in the split-bf16 modes -- out = in . W + bias with BatchNorm
    partial sums, and the accumulating form -- against float64; ragged last tile, 32-channel tail chunk (K = 96), one and several
    column tiles, more tiles than workgroups (the persistent walk), padded row strides, the four-slot weight ring of the 128-column form
    across a one-step chunk, a single step per tile and a partial column tile; the one-tile-per-workgroup control (tuning
    key 8) must give the same bits."""
    from fusion_gcn_amd import _lib, ops
    if not ops.pw_gemm_available():
        pytest.skip("pw_gemm runs in the split-bf16 math modes")
    x = rnd(rows, 1, 1, ld_in, seed=200)
    w = rnd(1, K, N, seed=201, scale=K ** -0.5)
    bias = rnd(N, seed=202)
    base = rnd(rows, 1, 1, ld_out, seed=203)
    want = x[..., :K].reshape(rows, K) @ w[0] + bias
    w3 = ops.pack_conv(to_gpu(w))                       # pack_split3, or the block-scaled two-way f16 split in math mode f16x2
    out = to_gpu(base)
    part = ops.pw_gemm(to_gpu(x), w3, out, bias=to_gpu(bias), stats=True)
    got = out.view(rows, ld_out).cpu().double()
    assert rel_l2(got[:, :N].numpy(), want.numpy()) < FWD_TOL
    assert torch.equal(got[:, N:], base.float().double().view(rows, ld_out)[:, N:])          # columns beyond N untouched
    tot = part.double().sum(0).cpu()
    assert rel_l2(tot[0].numpy(), want.sum(0).numpy()) < RED_TOL * 5 + 1e-7 * rows ** 0.5
    assert rel_l2(tot[1].numpy(), (want ** 2).sum(0).numpy()) < RED_TOL
    _lib.load().fgcn_set_tuning(8, 1)
    try:
        out1 = to_gpu(base)
        part1 = ops.pw_gemm(to_gpu(x), w3, out1, bias=to_gpu(bias), stats=True)
    finally:
        _lib.load().fgcn_set_tuning(8, 0)
    assert torch.equal(out1, out) and torch.equal(part1, part)
    acc = to_gpu(base)
    ops.pw_gemm(to_gpu(x), w3, acc, accumulate=True)
    want_acc = base.view(rows, ld_out)[:, :N] + x[..., :K].reshape(rows, K) @ w[0]
    assert rel_l2(acc.view(rows, ld_out)[:, :N].cpu().numpy(), want_acc.numpy()) < FWD_TOL
    with ops.math_mode("bf16"):
        outb = to_gpu(base)
        ops.pw_gemm(to_gpu(x), ops.pack_conv(to_gpu(w)), outb, bias=to_gpu(bias))
        assert rel_l2(outb.view(rows, ld_out)[:, :N].cpu().numpy(), want.numpy()) < 2e-2


@pytest.mark.math_modes("bf16x3")
@pytest.mark.parametrize("B,T,V,C", [(3, 37, 25, 64), (2, 21, 25, 128), (2, 9, 27, 256), (1, 50, 22, 64), (5, 3, 18, 128), (2, 40, 32, 64)])
def test_halo_temporal_conv_with_the_input_stage_fused(B, T, V, C):
    """North-star kernel 2 as the north star states it: G = relu(BatchNorm(y) + x) (agcn.py:113-115) formed INSIDE the 9x1 temporal
    conv while it stages its image (`fuse_in`), against the two-pass form (bn_act, then the conv on G): the conv output and its
    BatchNorm partial sums, G and the sign image written as by-products -- all bit for bit, they are the same arithmetic -- plus
    the conv output against the float64 formula.  Tiles straddle samples; T < the 8-frame halo is included."""
    from fusion_gcn_amd import ops
    if not ops.tconv_halo_bn_sums() or ops.get_math_mode() == "f16x2":
        pytest.skip("the fused input stage is built for the split-bf16 halo kernel (bf16x3 / bf16 products)")
    kt = 9
    wt = rnd(kt, C, C, seed=180, scale=(kt * C) ** -0.5)
    bias = rnd(C, seed=181)
    y, x = rnd(B, T, V, C, seed=182), rnd(B, T, V, C, seed=183)
    vec = torch.stack([rnd(C, seed=184), rnd(C, seed=185).abs() + 0.5, rnd(C, seed=186), rnd(C, seed=187) * 0.3])   # mean, rstd, scale, shift
    yg, xg, vg = to_gpu(y), to_gpu(x), to_gpu(vec)
    g64 = torch.relu(y.double() * vec[2].double() + vec[3].double() + x.double())
    want = ref_rows_conv(g64, wt, ops.conv_tmap(kt, 1), T, bias)

    def both_forms(tol):
        w4 = ops.pack_conv(to_gpu(wt))                   # the streamed form of the CURRENT math mode
        g_ref, sign_ref = ops.bn_act(yg, vg, xg, None, relu=True, sign_mask=True)
        u_ref = torch.empty(B, T, V, C, device=dev())
        part_ref = ops.tconv_halo(g_ref, w4, u_ref, Th=T, taps=kt, tb=1, tc=-4, bias=to_gpu(bias), stats=True)
        g = torch.full((B, T, V, C), 7.0, device=dev())
        sign = torch.full((B * T * V * C // 8,), 0xAA, device=dev(), dtype=torch.uint8)
        u = torch.full((B, T, V, C), 3.0, device=dev())
        part = ops.tconv_halo(yg, w4, u, Th=T, taps=kt, tb=1, tc=-4, bias=to_gpu(bias), stats=True, fuse_in=(vg, xg, g, sign))
        assert torch.equal(g, g_ref) and torch.equal(sign, sign_ref)
        assert torch.equal(u, u_ref)
        # (the partial sums are per row tile, and the two-pass form may run another row tiling -- 192-row tiles at 64 columns: their
        # totals agree to rounding)
        assert rel_l2(part.double().sum(0).cpu().numpy(), part_ref.double().sum(0).cpu().numpy()) < 1e-6
        assert rel_l2(u.cpu().numpy(), want.numpy()) < tol

    both_forms(FWD_TOL)
    with ops.math_mode("bf16"):                          # BASELINE config 5: the same kernel with one bf16 part per operand
        both_forms(2e-2)


@pytest.mark.parametrize("B,T,V,K,N,kt,s", [(2, 40, 25, 64, 64, 9, 1), (2, 41, 25, 64, 128, 9, 2),
                                            (4, 64, 25, 128, 96, 1, 1), (2, 30, 22, 4, 64, 1, 1),
                                            (1, 20, 18, 192, 64, 1, 1), (2, 33, 25, 64, 128, 1, 2)])
def test_rows_wgrad(B, T, V, K, N, kt, s):
    from fusion_gcn_amd import ops
    tmap = ops.conv_tmap(kt, s)
    T_out = (T - 1) // s + 1
    a = rnd(B, T, V, K, seed=10)
    g = rnd(B, T_out, V, N, seed=11)
    w = torch.zeros(kt, K, N, dtype=torch.float64, requires_grad=True)
    y = ref_rows_conv(a, w, tmap, T_out)
    (want,) = torch.autograd.grad((y * g).sum(), w)
    got = ops.rows_wgrad(to_gpu(a), to_gpu(g), K=K, N=N, tmap=tmap)
    assert rel_l2(got.cpu().numpy(), want.numpy()) < RED_TOL
    generic = ops.rows_wgrad(to_gpu(a), to_gpu(g), K=K, N=N, tmap=tmap, wide=False)
    assert rel_l2(generic.cpu().numpy(), want.numpy()) < RED_TOL


@pytest.mark.parametrize("B,T,V,K,N,s", [
    (2, 30, 25, 192, 64, 1),      # conv_d on the stacked agg, 64 columns: 2 tiles of 3 chunks, two row halves per stage
    (1, 12, 25, 768, 256, 1),     # 24 chunks = 4 tiles of 6, two column tiles
    (3, 20, 18, 64, 96, 1),       # theta|phi embedding: N tail inside a 128-column tile
    (2, 21, 27, 64, 128, 2),      # strided residual conv (frames 2t of a), odd T
    (2, 9, 20, 160, 36, 1),       # 5 chunks, ragged N
    (1, 1, 5, 32, 4, 1),          # tiny: one frame, one chunk
    (2, 10, 25, 224, 64, 1),      # 7 chunks: not a multiple of the per-wave chunk count (tile tail reads zeros)
])
def test_pointwise_wgrad_channel_chunks(B, T, V, K, N, s):
    """1x1 weight gradients on the multi-accumulator kernel (one accumulator per 32-channel chunk) equal autograd and the
    generic kernel; channel windows (a_coff / g_coff) and accumulation work; runs are bitwise reproducible."""
    from fusion_gcn_amd import ops
    T_out = (T - 1) // s + 1
    tmap = (1, s, 0, 0, 1)
    a, g = rnd(B, T, V, K + 32, seed=41), rnd(B, T_out, V, N + 8, seed=42)
    w = torch.zeros(1, K, N, dtype=torch.float64, requires_grad=True)
    y = ref_rows_conv(a[..., 32:], w, tmap, T_out)
    (want,) = torch.autograd.grad((y * g[..., 4:4 + N]).sum(), w)
    kw = dict(K=K, N=N, tmap=tmap, a_coff=32, g_coff=4)
    got = ops.rows_wgrad(to_gpu(a), to_gpu(g), wide=True, **kw)
    assert rel_l2(got.cpu().numpy(), want.numpy()) < RED_TOL
    ref = ops.rows_wgrad(to_gpu(a), to_gpu(g), wide=False, **kw)
    assert rel_l2(got.cpu().numpy(), ref.cpu().numpy()) < RED_TOL
    assert torch.equal(got, ops.rows_wgrad(to_gpu(a), to_gpu(g), wide=True, **kw))
    acc = got.clone()
    ops.rows_wgrad(to_gpu(a), to_gpu(g), wide=True, out=acc, accumulate=True, **kw)
    assert rel_l2(acc.cpu().numpy(), 2 * want.numpy()) < RED_TOL


def test_weight_gradients_in_parameter_layout():
    """conv_param=(groups, K_true): the slab reduction writes the conv parameter's own (out, in, taps, 1) layout, dropping
    padded input channels and splitting stacked convolutions (conv_d's three subsets)."""
    from fusion_gcn_amd import ops
    B, T, V = 2, 12, 25
    a, g = rnd(B, T, V, 64, seed=51), rnd(B, T, V, 96, seed=52)
    ref = ops.tconv_wgrad(to_gpu(a), to_gpu(g), taps=9)                              # (9, 64, 96)
    got = ops.tconv_wgrad(to_gpu(a), to_gpu(g), taps=9, conv_param=(1, 64))          # (96, 64, 9, 1)
    assert got.shape == (96, 64, 9, 1) and got.is_contiguous()
    assert torch.equal(got[..., 0], ref.permute(2, 1, 0))
    ref1 = ops.rows_wgrad(to_gpu(a), to_gpu(g), K=64, N=96)[0]                       # (64, 96)
    got1 = ops.rows_wgrad(to_gpu(a), to_gpu(g), K=64, N=96, conv_param=(1, 61))      # 3 padded input channels dropped
    assert got1.shape == (96, 61, 1, 1) and torch.equal(got1[:, :, 0, 0], ref1[:61].t())
    a3 = rnd(B, T, V, 3 * 128, seed=53)
    for wide in (False, True):
        ref3 = ops.rows_wgrad(to_gpu(a3), to_gpu(g), K=384, N=96, wide=wide)[0]      # (3*128, 96)
        got3 = ops.rows_wgrad(to_gpu(a3), to_gpu(g), K=384, N=96, wide=wide, conv_param=(3, 128))
        assert got3.shape == (3, 96, 128, 1, 1)
        for k in range(3):
            assert torch.equal(got3[k, :, :, 0, 0], ref3[128 * k:128 * (k + 1)].t())
    acc = got1.clone()
    ops.rows_wgrad(to_gpu(a), to_gpu(g), K=64, N=96, conv_param=(1, 61), out=acc, accumulate=True)
    assert rel_l2(acc.cpu().numpy(), 2 * got1.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("B,V,K,N", [(8, 652, 652, 64), (3, 100, 16, 100), (2, 300, 128, 128), (5, 36, 36, 8)])
def test_batched_row_gemms_one_and_two_levels(B, V, K, N):
    """fgcn_rows_gemm_batched / _batched2 (the per-sample products of the IMU AGCN convolution, graph_convolution.py:96-101): B x 3
    problems out[b, :, k] = a[b, k] . w[k, b] with the operands laid out (sample, subset) and (subset, sample), one launch, against
    float64; the small-problem tile rule against the large-problem tiles (tuning key 24) to rounding; accumulate."""
    from fusion_gcn_amd import _lib, ops
    lib = _lib.load()
    a = rnd(B, 3, V, K, seed=30, scale=0.3)                                      # (sample, subset, rows, K)
    w = rnd(3, B, K, N, seed=31, scale=0.3)                                      # (subset, sample, K, N)
    want = torch.einsum("bkvc,kbcn->bvkn", a, w).reshape(B, V, 3 * N)            # (sample, rows, subset * N)
    ag, wg = to_gpu(a), to_gpu(w)

    def run():
        out = torch.zeros(B, V, 3 * N, device=dev())
        ops.rows_gemm_batched(ag, wg, out, batch=B, rows=V, K=K, N=N, ld_in=K, ld_out=3 * N, in_bs=3 * V * K, w_bs=K * N,
                              out_bs=V * 3 * N, inner=3, in_bs2=V * K, w_bs2=B * K * N, out_bs2=N)
        return out
    got = run()
    assert rel_l2(got.cpu().numpy(), want.numpy()) < FWD_TOL
    assert torch.equal(got, run())
    lib.fgcn_set_tuning(24, 1)
    try:
        big = run()
    finally:
        lib.fgcn_set_tuning(24, 0)
    assert rel_l2(big.cpu().numpy(), got.cpu().numpy()) < 1e-6
    one = torch.zeros_like(got)                                                  # the same, one launch per subset
    for k in range(3):
        ops.rows_gemm_batched(ag, wg, one, batch=B, rows=V, K=K, N=N, ld_in=K, ld_out=3 * N, in_bs=3 * V * K, w_bs=K * N,
                              out_bs=V * 3 * N, in_off=k * V * K, w_off=k * B * K * N, out_off=k * N)
    assert torch.equal(one, got)
    # more problems than one launch's grid.z carries go out in chunks of the outer level (here: forced to 2 samples per launch)
    limit = ops.ROWS_GEMM_MAX_PROBLEMS
    ops.ROWS_GEMM_MAX_PROBLEMS = 6
    try:
        assert rel_l2(run().cpu().numpy(), got.cpu().numpy()) < 1e-6      # (the tile rule may pick other tiles for a smaller launch)
    finally:
        ops.ROWS_GEMM_MAX_PROBLEMS = limit
    ops.rows_gemm_batched(ag, wg, got, batch=B, rows=V, K=K, N=N, ld_in=K, ld_out=3 * N, in_bs=3 * V * K, w_bs=K * N,
                          out_bs=V * 3 * N, inner=3, in_bs2=V * K, w_bs2=B * K * N, out_bs2=N, accumulate=True)
    assert rel_l2(got.cpu().numpy(), 2 * want.numpy()) < FWD_TOL
    with pytest.raises(_lib.FgcnError):
        ops.rows_gemm_batched(ag, wg, got, batch=B, rows=V, K=K, N=N, ld_in=K, ld_out=3 * N, in_bs=3 * V * K, w_bs=K * N,
                              out_bs=V * 3 * N, inner=4, in_bs2=V * K, w_bs2=B * K * N, out_bs2=N)


def test_reduce_sum_and_col_sum_and_pack():
    from fusion_gcn_amd import ops
    src = rnd(37, 1000, seed=12)
    dst = torch.ones(1000, device=dev())
    ops.reduce_sum(to_gpu(src), dst, accumulate=True)
    assert rel_l2(dst.cpu().numpy(), (src.sum(0) + 1).numpy()) < 1e-6
    x = rnd(3, 50, 25, 96, seed=13)
    assert rel_l2(ops.col_sum(to_gpu(x), 96).cpu().numpy(), x.reshape(-1, 96).sum(0).numpy()) < RED_TOL
    assert rel_l2(ops.col_sum(to_gpu(x), 32, coff=64).cpu().numpy(), x.reshape(-1, 96)[:, 64:].sum(0).numpy()) < RED_TOL
    w = rnd(64, 32, 9, 1, seed=14)                                         # conv weight (O, C, kt, 1)
    packed = ops.pack_weight(to_gpu(w), taps=9, K=32, N=64, st_tap=1, st_k=9, st_n=32 * 9)
    np.testing.assert_array_equal(packed.cpu().numpy(), w[..., 0].permute(2, 1, 0).float().numpy())
    flipped = ops.pack_weight(to_gpu(w), taps=9, K=32, N=62, st_tap=1, st_k=9, st_n=32 * 9, n_pad=64, flip=True)
    want = torch.zeros(9, 32, 64)
    want[..., :62] = w[:62, :, :, 0].permute(2, 1, 0).flip(0).float()
    np.testing.assert_array_equal(flipped.cpu().numpy(), want.numpy())


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("V,cin", [(25, 64), (18, 128), (22, 4), (27, 256), (20, 64)])
def test_joint_mix_aggregation_and_its_transpose(V, cin):
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.block import spec_agg, spec_dx
    B, T = 3, 9
    x = rnd(B, T, V, cin, seed=20)
    a = rnd(B, 3, V, V, seed=21, scale=0.3)
    want = torch.einsum("btvc,bkvw->btwkc", x, a).reshape(B, T, V, 3 * cin)
    agg = torch.full((B, T, V, 3 * cin), 7.0, device=dev())
    ops.joint_mix(to_gpu(x), agg, to_gpu(a), spec_agg(cin), in_channels=cin, out_channels=3 * cin)
    assert rel_l2(agg.cpu().numpy(), want.numpy()) < FWD_TOL
    # dx[v, c] = sum_k sum_w dagg[w, (k, c)] a[k, v, w], accumulated onto an existing tensor
    dagg = rnd(B, T, V, 3 * cin, seed=22)
    base = rnd(B, T, V, cin, seed=23)
    want_dx = base + torch.einsum("btwkc,bkvw->btvc", dagg.reshape(B, T, V, 3, cin), a)
    dx = to_gpu(base)
    ops.joint_mix(to_gpu(dagg), dx, to_gpu(a), spec_dx(cin), in_channels=3 * cin, out_channels=cin, accumulate=True)
    assert rel_l2(dx.cpu().numpy(), want_dx.numpy()) < FWD_TOL
    # shared (static) adjacency
    ops.joint_mix(to_gpu(x), agg, to_gpu(a[:1]), spec_agg(cin), in_channels=cin, out_channels=3 * cin)
    want = torch.einsum("btvc,kvw->btwkc", x, a[0]).reshape(B, T, V, 3 * cin)
    assert rel_l2(agg.cpu().numpy(), want.numpy()) < FWD_TOL


@pytest.mark.parametrize("order", [(1,), (2, 1)])
@pytest.mark.parametrize("V,cin", [(25, 64), (18, 128), (27, 256), (22, 192), (25, 4), (20, 32), (32, 64), (5, 8)])
def test_joint_mix_channel_groups(V, cin, order, monkeypatch):
    """The channel-group kernel (1 or 2 channels per lane; every k-step template) gives the same agg / dx as the
    einsum, with and without accumulation, and never writes outside its rows (canary row stride)."""
    from fusion_gcn_amd import block, ops
    monkeypatch.setattr(ops.paths(), "mix_vw_order", order)      # (a per-context path option: fusion_gcn_amd/paths.py)
    B, T = 3, 9
    x, a = rnd(B, T, V, cin, seed=70), rnd(B, 3, V, V, seed=71, scale=0.3)
    want = torch.einsum("btvc,bkvw->btwkc", x, a).reshape(B, T, V, 3 * cin)
    agg = torch.full((B, T, V, 3 * cin), 7.0, device=dev())
    block.mix_agg(to_gpu(x), agg, to_gpu(a), cin)
    assert rel_l2(agg.cpu().numpy(), want.numpy()) < FWD_TOL
    dagg, base = rnd(B, T, V, 3 * cin, seed=72), rnd(B, T, V, cin, seed=73)
    want_dx = base + torch.einsum("btwkc,bkvw->btvc", dagg.reshape(B, T, V, 3, cin), a)
    dx = to_gpu(base)
    block.mix_dx(to_gpu(dagg), dx, to_gpu(a), cin, accumulate=True)
    assert rel_l2(dx.cpu().numpy(), want_dx.numpy()) < FWD_TOL
    block.mix_dx(to_gpu(dagg), dx, to_gpu(a), cin, accumulate=False)
    assert rel_l2(dx.cpu().numpy(), (want_dx - base).numpy()) < FWD_TOL
    # static (shared) adjacency: one set of matrices for the whole batch
    block.mix_agg(to_gpu(x), agg, to_gpu(a[:1]), cin)
    want1 = torch.einsum("btvc,kvw->btwkc", x, a[0]).reshape(B, T, V, 3 * cin)
    assert rel_l2(agg.cpu().numpy(), want1.numpy()) < FWD_TOL


@pytest.mark.parametrize("order", [None, (1,), (2, 1)])
@pytest.mark.parametrize("ic", [16, 32, 64, 8, 48])
def test_joint_mix_embedding_gradient(ic, order, monkeypatch):
    from fusion_gcn_amd import block, ops
    B, T, V = 2, 7, 25
    emb = rnd(B, T, V, 6 * ic, seed=24)
    ds = rnd(B, 3, V, V, seed=25)
    e = emb.reshape(B, T, V, 3, 2, ic)
    theta, phi = e[..., 0, :], e[..., 1, :]                                  # (B,T,V,3,ic)
    dtheta = torch.einsum("bkvw,btwke->btvke", ds, phi)
    dphi = torch.einsum("bkvw,btvke->btwke", ds, theta)
    want = torch.stack([dtheta, dphi], dim=4).reshape(B, T, V, 6 * ic)
    out = torch.full((B, T, V, 6 * ic), 3.0, device=dev())
    if order is None:       # the dword kernel with 16-lane masks (kept for widths the channel-group kernel cannot tile)
        if ic % 16:
            pytest.skip("dword-kernel spec needs 16-channel groups")
        ops.joint_mix(to_gpu(emb), out, to_gpu(ds), block.spec_demb(ic), in_channels=6 * ic, out_channels=6 * ic)
    else:
        monkeypatch.setattr(ops.paths(), "mix_vw_order", order)      # (a per-context path option: fusion_gcn_amd/paths.py)
        sums = block.mix_demb(to_gpu(emb), out, to_gpu(ds), ic)
        # the column sums (theta|phi bias gradient) come out of the same launch
        assert rel_l2(sums.cpu().numpy(), want.reshape(-1, 6 * ic).sum(0).numpy()) < RED_TOL
    assert rel_l2(out.cpu().numpy(), want.numpy()) < FWD_TOL


@pytest.mark.parametrize("V,T,cin,cout,B", [(25, 30, 64, 64, 3), (18, 33, 64, 128, 2), (27, 12, 128, 128, 2), (22, 9, 4, 64, 2),
                                            (32, 5, 96, 36, 1), (25, 300, 64, 64, 2), (25, 7, 256, 256, 1)])
def test_spatial_wgrad_fused_agg_recompute(V, T, cin, cout, B):
    """conv_d weight gradient with agg = x . A^ formed on chip equals the float64 einsum and the unfused pair
    (joint_mix_vec + row weight gradient), in both output layouts; bitwise reproducible."""
    from fusion_gcn_amd import block, ops
    x, a, dy = rnd(B, T, V, cin, seed=95), rnd(B, 3, V, V, seed=96, scale=0.3), rnd(B, T, V, cout, seed=97)
    agg = torch.einsum("btvc,bkvw->btwkc", x, a)                              # (B,T,W,3,cin)
    want = torch.einsum("btwkc,btwo->kco", agg, dy).reshape(3 * cin, cout)
    got = ops.spatial_wgrad(to_gpu(x), to_gpu(dy), to_gpu(a))
    assert got.shape == (1, 3 * cin, cout)
    assert rel_l2(got[0].cpu().numpy(), want.numpy()) < RED_TOL
    assert torch.equal(got, ops.spatial_wgrad(to_gpu(x), to_gpu(dy), to_gpu(a)))
    agg_g = torch.empty(B, T, V, 3 * cin, device=dev())
    block.mix_agg(to_gpu(x), agg_g, to_gpu(a), cin)
    ref = ops.rows_wgrad(agg_g, to_gpu(dy), K=3 * cin, N=cout)
    assert rel_l2(got.cpu().numpy(), ref.cpu().numpy()) < RED_TOL
    pl = ops.spatial_wgrad(to_gpu(x), to_gpu(dy), to_gpu(a), conv_param=(3, cin - (1 if cin == 4 else 0)))
    k_true = cin - (1 if cin == 4 else 0)
    assert pl.shape == (3, cout, k_true, 1, 1)
    for k in range(3):
        assert torch.equal(pl[k, :, :, 0, 0], got[0, k * cin:k * cin + k_true].t())
    shared = ops.spatial_wgrad(to_gpu(x), to_gpu(dy), to_gpu(a[:1]))         # static adjacency: one matrix set
    want1 = torch.einsum("btwkc,btwo->kco", torch.einsum("btvc,kvw->btwkc", x, a[0]), dy).reshape(3 * cin, cout)
    assert rel_l2(shared[0].cpu().numpy(), want1.numpy()) < RED_TOL


@pytest.mark.parametrize("rows,C,n", [(1000, 16, 6), (777, 32, 6), (4100, 64, 3), (130, 4, 2)])
def test_batchnorm_into_channel_windows(rows, C, n):
    """ops.bn_apply_window / ops.bn_bwd_window (fgcn_bn_apply_ld, fgcn_bn_bwd_reduce_ld / _apply_ld): n plain BatchNorms whose results are
    the channel windows of one wide tensor (MS-G3D's branch concatenation, ms_tcn.py:88-109) against torch: forward = cat of the
    normalised branches, backward from the wide gradient read in place = the per-branch BatchNorm backward of its slice."""
    from fusion_gcn_amd import ops
    torch.manual_seed(7)
    a = [rnd(1, rows, 1, C, seed=500 + i, scale=1.0 + 0.3 * i) + 0.2 * i for i in range(n)]
    gamma = [rnd(C, seed=520 + i).abs() + 0.5 for i in range(n)]
    beta = [rnd(C, seed=540 + i) for i in range(n)]
    dwide = rnd(1, rows, 1, n * C, seed=560)
    out = torch.full((1, rows, 1, n * C), 7.0, device=dev())
    vecs = []
    for i in range(n):
        ag = to_gpu(a[i])
        part = ops.col_moments(ag)
        vec = ops.bn_finalize(part, rows, to_gpu(gamma[i]), to_gpu(beta[i]), None, None)
        ops.bn_apply_window(ag, vec, out, i * C)
        vecs.append(vec)
    want_parts, leaves = [], []
    for i in range(n):
        x = a[i].double().requires_grad_(True)
        g, b = gamma[i].double().requires_grad_(True), beta[i].double().requires_grad_(True)
        mean, var = x.mean((0, 1, 2)), x.var((0, 1, 2), unbiased=False)
        want_parts.append((x - mean) / torch.sqrt(var + 1e-5) * g + b)
        leaves.append((x, g, b))
    want = torch.cat(want_parts, dim=-1)
    assert rel_l2(out.cpu().numpy(), want.detach().numpy()) < FWD_TOL
    want.backward(dwide.double())
    dw = to_gpu(dwide)
    for i in range(n):
        da, sums = ops.bn_bwd_window(dw, i * C, to_gpu(a[i]), vecs[i], train=True)
        x, g, b = leaves[i]
        assert rel_l2(da.cpu().numpy(), x.grad.numpy()) < 5 * FWD_TOL
        assert rel_l2(sums[1].cpu().numpy(), g.grad.numpy()) < RED_TOL * 5
        assert rel_l2(sums[0].cpu().numpy(), b.grad.numpy()) < RED_TOL * 5


@pytest.mark.math_modes("bf16x3", "f16x2")
@pytest.mark.parametrize("B,T,V,C", [(3, 37, 25, 64), (2, 20, 27, 64), (2, 9, 25, 128), (1, 5, 28, 64)])
def test_halo_conv_wave_arrangements_agree(B, T, V, C):
    """The nine-tap halo conv as 4 x 1 waves over 192-row tiles (64 output columns: used from 1536 tiles on, forced here by tuning key 7
    bit 6; one 128-column tile: the default for 64 < N <= 128) against 2 x 2 waves over 128-row tiles (key 7 bits 3 / 5): the outputs
    bit for bit (each output element is the same sum in the same order), the BatchNorm partial sums in total; forward, data gradient
    with the BatchNorm-backward sums in its epilogue, and the accumulating second pass of a strided forward."""
    from fusion_gcn_amd import _lib, ops
    if not ops.tconv_halo_bn_sums():
        pytest.skip("the split-bf16 halo kernel")
    lib = _lib.load()
    kt = 9
    wt = rnd(kt, C, C, seed=400, scale=(kt * C) ** -0.5)
    g, bias = to_gpu(rnd(B, T, V, C, seed=401)), to_gpu(rnd(C, seed=402))
    y = to_gpu(rnd(B, T, V, C, seed=403))
    vec = to_gpu(torch.stack([rnd(C, seed=404), rnd(C, seed=405).abs() + 0.5, rnd(C, seed=406), rnd(C, seed=407)]))
    _, sign = ops.bn_act(y, vec, g, None, relu=True, sign_mask=True)

    def run(bn_bwd):
        w4 = ops.pack_conv(to_gpu(wt))
        u = torch.full((B, T, V, C), 3.0, device=dev())
        part = ops.tconv_halo(g, w4, u, Th=T, taps=kt, tb=1, tc=-4, bias=bias, stats=not bn_bwd, bn_bwd=(y, sign, vec) if bn_bwd else None)
        acc = u.clone()
        ops.tconv_halo(g, w4, acc, Th=T, taps=kt, tb=1, tc=-4, accumulate=True, stats=True)
        return u, part, acc
    try:
        outs = {}
        for name, key in (("2x2", 8 | 32), ("4x1", 64)):
            lib.fgcn_set_tuning(7, key)
            outs[name] = [run(False), run(True) if C <= 64 else None]
    finally:
        lib.fgcn_set_tuning(7, 0)
    for a, b in zip(outs["2x2"], outs["4x1"]):
        if a is None:
            continue
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
        assert rel_l2(a[1].double().sum(0).cpu().numpy(), b[1].double().sum(0).cpu().numpy()) < 1e-6


@pytest.mark.math_modes("bf16x3")
@pytest.mark.parametrize("V,T,cin,cout,B", [(25, 13, 64, 64, 2), (25, 7, 128, 256, 2), (27, 9, 64, 128, 1), (18, 10, 64, 64, 2),
                                             (16, 8, 128, 128, 1), (32, 5, 64, 96, 1), (22, 31, 256, 256, 1), (25, 300, 64, 64, 1)])
def test_spatial_forward_tile_form(V, T, cin, cout, B):
    """The fused spatial forward in its tile form (fgcn_spatial_tile.hip: 128 / V whole frames per workgroup, the aggregation of a
    (channel tile, subset) pair formed once per workgroup into an LDS image): y = sum_k (x . A^_k) . Wd_k + bias against the float64
    einsums, its BatchNorm partial sums against the output it wrote, per-sample and shared adjacency, ragged last frame group
    (T % F != 0), 4 .. 8 frames per tile, a partial column tile; bitwise reproducible; agrees with the two-frames-per-wave form."""
    from fusion_gcn_amd import ops
    if not ops.spatial_fwd_tile_available(V, cin, cout):
        pytest.skip("the tile form runs with the bf16x3 products")
    x, a = rnd(B, T, V, cin, seed=300), rnd(B, 3, V, V, seed=301, scale=0.3)
    wd, bias = rnd(3, cin, cout, seed=302, scale=(3 * cin) ** -0.5), rnd(cout, seed=303)
    want = torch.einsum("btwkc,kco->btwo", torch.einsum("btvc,bkvw->btwkc", x.double(), a.double()), wd.double()) + bias.double()
    w3 = ops.pack_split3(to_gpu(wd.reshape(1, 3 * cin, cout)))
    y, part = ops.spatial_fwd_tile(to_gpu(x), to_gpu(a), w3, to_gpu(bias), Cin=cin, Cout=cout, stats=True)
    assert rel_l2(y.cpu().numpy(), want.numpy()) < FWD_TOL
    s = part.double().sum(0).cpu()
    assert rel_l2(s[0].numpy(), y.double().sum((0, 1, 2)).cpu().numpy()) < RED_TOL
    assert rel_l2(s[1].numpy(), (y.double() ** 2).sum((0, 1, 2)).cpu().numpy()) < RED_TOL
    y2, part2 = ops.spatial_fwd_tile(to_gpu(x), to_gpu(a), w3, to_gpu(bias), Cin=cin, Cout=cout, stats=True)
    assert torch.equal(y, y2) and torch.equal(part, part2)
    y_old, _ = ops.spatial_fwd(to_gpu(x), to_gpu(a), ops.pack_spatial(to_gpu(wd.reshape(3 * cin, cout)), cin), to_gpu(bias), Cin=cin,
                               Cout=cout, stats=False)
    assert rel_l2(y.cpu().numpy(), y_old.cpu().numpy()) < FWD_TOL
    shared, _ = ops.spatial_fwd_tile(to_gpu(x), to_gpu(a[:1]), w3, None, Cin=cin, Cout=cout, stats=False)
    want1 = torch.einsum("btwkc,kco->btwo", torch.einsum("btvc,kvw->btwkc", x.double(), a[0].double()), wd.double())
    assert rel_l2(shared.cpu().numpy(), want1.numpy()) < FWD_TOL


@pytest.mark.parametrize("V,T,C,B", [(25, 30, 64, 3), (18, 33, 128, 2), (27, 12, 256, 2), (22, 9, 4, 2), (32, 5, 96, 1), (25, 300, 64, 2)])
def test_joint_dagg_fused_dx_and_gram(V, T, C, B):
    """One pass over dagg gives both dx (+)= sum_k dagg_k . A^_k^T and dA^_k = x^T dagg_k (float64 einsums), with and without
    accumulation, per-sample and shared matrices; the partial grams sum to what joint_gram produces."""
    from fusion_gcn_amd import ops
    x, a = rnd(B, T, V, C, seed=90), rnd(B, 3, V, V, seed=91, scale=0.3)
    dagg, base = rnd(B, T, V, 3 * C, seed=92), rnd(B, T, V, C, seed=93)
    d4 = dagg.reshape(B, T, V, 3, C)
    want_dx = torch.einsum("btwkc,bkvw->btvc", d4, a)
    want_g = torch.einsum("btvc,btwkc->bkvw", x, d4)
    for acc in (False, True):
        dx = to_gpu(base)
        part = ops.joint_dagg(to_gpu(x), to_gpu(dagg), to_gpu(a), dx, accumulate=acc)
        assert rel_l2(dx.cpu().numpy(), (want_dx + (base if acc else 0)).numpy()) < FWD_TOL
        got_g = part.double().sum(1)[:, :, :V, :V].cpu()
        assert rel_l2(got_g.numpy(), want_g.numpy()) < RED_TOL
        assert float(part[:, :, :, V:, :].abs().max() if V < 32 else 0.0) == 0.0        # padding stays zero
    ref = ops.joint_gram(to_gpu(x), to_gpu(dagg), [(0, k * C, C) for k in range(3)])
    assert rel_l2(part.sum(1).cpu().numpy(), ref.sum(1).cpu().numpy()) < RED_TOL
    dx = to_gpu(base)
    ops.joint_dagg(to_gpu(x), to_gpu(dagg), to_gpu(a[:1]), dx, accumulate=False)     # static (shared) adjacency
    assert rel_l2(dx.cpu().numpy(), torch.einsum("btwkc,kvw->btvc", d4, a[0]).numpy()) < FWD_TOL


@pytest.mark.math_modes("bf16x3", "f16x2")
@pytest.mark.parametrize("V,T,cin,cout,B", [(25, 13, 64, 64, 2), (25, 7, 128, 256, 2), (27, 9, 64, 128, 1), (18, 10, 64, 64, 2),
                                             (16, 8, 128, 128, 1), (32, 5, 64, 64, 1), (22, 31, 256, 256, 1), (25, 300, 64, 64, 1),
                                             (17, 23, 64, 128, 2), (21, 12, 128, 64, 3)])
def test_spatial_backward_tile_form(V, T, cin, cout, B):
    """The fused backward of the spatial stage in tile form (fgcn_spatial_bwd_tile.hip: dagg = dy . Wd on chip only): dx (+)=
    sum_k dagg_k . A^_k^T and the partial grams dA^_k = x^T dagg_k against the float64 einsums (backward of agcn.py:103-111), with and
    without accumulation, per-sample and shared adjacency, ragged last frame group (T % F != 0), 4 .. 8 frames per tile, several
    64-channel input groups; bitwise reproducible; agrees with the unfused pair pw_gemm + joint_dagg."""
    from fusion_gcn_amd import ops
    if not ops.spatial_bwd_tile_available(V, cin, cout):
        pytest.skip("the tile form runs with the bf16x3 products")
    x, a = rnd(B, T, V, cin, seed=310), rnd(B, 3, V, V, seed=311, scale=0.3)
    dy, base = rnd(B, T, V, cout, seed=312), rnd(B, T, V, cin, seed=313)
    wd = rnd(3, cout, cin, seed=314, scale=cout ** -0.5)                       # Wd_k[o][c]
    dagg = torch.einsum("btwo,koc->btwkc", dy, wd)
    want_dx = torch.einsum("btwkc,bkvw->btvc", dagg, a)
    want_g = torch.einsum("btvc,btwkc->bkvw", x, dagg)
    wt = wd.permute(1, 0, 2).reshape(1, cout, 3 * cin)                         # [o][k cin + c]
    w3 = ops.pack_split3(to_gpu(wt))
    parts = []
    for acc in (False, True):
        dx = to_gpu(base)
        part = ops.spatial_bwd_tile(to_gpu(dy), to_gpu(x), to_gpu(a), w3, dx, accumulate=acc)
        assert rel_l2(dx.cpu().numpy(), (want_dx + (base if acc else 0)).numpy()) < FWD_TOL, acc
        got_g = part.double().sum(1)[:, :, :V, :V].cpu()
        assert rel_l2(got_g.numpy(), want_g.numpy()) < RED_TOL
        if V < 32:
            assert float(part[:, :, :, V:, :].abs().max()) == 0.0 and float(part[:, :, :, :, V:].abs().max()) == 0.0   # padding stays zero
        parts.append((dx, part))
    dx2 = to_gpu(base)
    part2 = ops.spatial_bwd_tile(to_gpu(dy), to_gpu(x), to_gpu(a), w3, dx2, accumulate=True)
    assert torch.equal(dx2, parts[1][0]) and torch.equal(part2, parts[1][1])
    # the unfused pair on the same data
    dagg_g = torch.empty(B, T, V, 3 * cin, device=dev())
    ops.pw_gemm(to_gpu(dy), w3, dagg_g)
    dx_old = to_gpu(base)
    part_old = ops.joint_dagg(to_gpu(x), dagg_g, to_gpu(a), dx_old, accumulate=True)
    assert rel_l2(dx2.cpu().numpy(), dx_old.cpu().numpy()) < FWD_TOL
    assert rel_l2(part2.sum(1).cpu().numpy()[:, :, :V, :V], part_old.sum(1).cpu().numpy()[:, :, :V, :V]) < RED_TOL
    dx = to_gpu(base)
    ops.spatial_bwd_tile(to_gpu(dy), to_gpu(x), to_gpu(a[:1]), w3, dx, accumulate=False)     # static (shared) adjacency
    assert rel_l2(dx.cpu().numpy(), torch.einsum("btwkc,kvw->btvc", dagg, a[0]).numpy()) < FWD_TOL
    # the ReLU-gated gradients of the block's two identity shortcuts (agcn.py:114,135) as the mix accumulators' start: dx = mix +
    # sum_i e_i * [bit of mask_i], the sign images from the real bn_act kernel
    ident = torch.stack([torch.zeros(cin), torch.ones(cin), torch.ones(cin), torch.zeros(cin)]).float().to(dev())
    want_gated, gated = want_dx.clone(), []
    for i in range(2):
        e, pre = rnd(B, T, V, cin, seed=320 + i), rnd(B, T, V, cin, seed=330 + i)
        _, mask = ops.bn_act(to_gpu(pre), ident, relu=True, sign_mask=True)
        want_gated = want_gated + e * (pre.float() > 0)
        gated.append((to_gpu(e), mask))
    dx = torch.full_like(to_gpu(base), float("nan"))                                         # every element is written
    part_g = ops.spatial_bwd_tile(to_gpu(dy), to_gpu(x), to_gpu(a), w3, dx, accumulate=False, gated=gated)
    assert rel_l2(dx.cpu().numpy(), want_gated.numpy()) < FWD_TOL
    assert torch.equal(part_g, parts[0][1])


@pytest.mark.math_modes("bf16x3", "f16x2")
@pytest.mark.parametrize("V,T,cin,cout,B", [(25, 13, 64, 64, 2), (25, 7, 128, 256, 2), (27, 9, 64, 128, 1), (18, 10, 64, 64, 2),
                                             (16, 8, 128, 128, 1), (32, 5, 64, 64, 1), (22, 31, 256, 256, 1), (25, 300, 64, 64, 1),
                                             (17, 23, 64, 128, 2), (21, 12, 128, 64, 3), (25, 20, 192, 64, 5), (19, 2, 64, 192, 3)])
def test_spatial_weight_gradient_tile_form(V, T, cin, cout, B):
    """conv_d's weight gradient in tile form (fgcn_spatial_wgrad_tile.hip: agg = x . A^ on chip only) against the float64 einsum
    (backward of agcn.py:103-111 with respect to conv_d[k].weight): per-sample and shared adjacency, ragged last frame group,
    4 .. 8 frames per tile, both wave arrangements (64 / 128 input channels per workgroup) and both tile widths, segments that span
    samples and segments shorter than a sample (tuning key 16), the parameter layout; bitwise reproducible; agrees with
    fgcn_spatial_wgrad."""
    from fusion_gcn_amd import _lib, ops
    if not ops.spatial_wgrad_tile_available(V, cin, cout):
        pytest.skip("the tile form runs with the bf16x3 products")
    x, a, dy = rnd(B, T, V, cin, seed=340), rnd(B, 3, V, V, seed=341, scale=0.3), rnd(B, T, V, cout, seed=342)
    want = torch.einsum("btvc,bkvw,btwo->kco", x.double(), a.double(), dy.double()).reshape(1, 3 * cin, cout)
    got = ops.spatial_wgrad_tile(to_gpu(x), to_gpu(dy), to_gpu(a))
    assert tuple(got.shape) == (1, 3 * cin, cout)
    assert rel_l2(got.cpu().numpy(), want.numpy()) < RED_TOL
    assert torch.equal(got, ops.spatial_wgrad_tile(to_gpu(x), to_gpu(dy), to_gpu(a)))
    old = ops.spatial_wgrad(to_gpu(x), to_gpu(dy), to_gpu(a))
    assert rel_l2(got.cpu().numpy(), old.cpu().numpy()) < RED_TOL
    lib = _lib.load()
    slabs = {0: lib.fgcn_spatial_wgrad_tile_slabs(B, T, V, cin, cout)}
    for target in (2, 100000):                  # one or two long segments (several samples each) / one (sample, tile) pair per workgroup
        try:
            assert lib.fgcn_set_tuning(16, target) == 0
            slabs[target] = lib.fgcn_spatial_wgrad_tile_slabs(B, T, V, cin, cout)
            g2 = ops.spatial_wgrad_tile(to_gpu(x), to_gpu(dy), to_gpu(a))
        finally:
            assert lib.fgcn_set_tuning(16, 0) == 0
        assert rel_l2(g2.cpu().numpy(), want.numpy()) < RED_TOL, target
    assert slabs[2] <= 2 and slabs[2] <= slabs[0] <= slabs[100000], slabs                  # the setting really changed the segmentation
    frames = min(8, 128 // V + 1)                                                          # frames per tile: (F - 1) V + 32 <= 160 rows
    assert slabs[100000] == B * ((T + frames - 1) // frames), slabs
    for tiles in (0, 2):                        # the widest tiles the channels allow (default) / 64 x 64 tiles
        try:
            assert lib.fgcn_set_tuning(21, tiles) == 0
            gt = ops.spatial_wgrad_tile(to_gpu(x), to_gpu(dy), to_gpu(a))
        finally:
            assert lib.fgcn_set_tuning(21, 0) == 0
        assert rel_l2(gt.cpu().numpy(), want.numpy()) < RED_TOL, tiles
    shared = ops.spatial_wgrad_tile(to_gpu(x), to_gpu(dy), to_gpu(a[:1]))                    # static (shared) adjacency
    want_s = torch.einsum("btvc,kvw,btwo->kco", x.double(), a[0].double(), dy.double()).reshape(1, 3 * cin, cout)
    assert rel_l2(shared.cpu().numpy(), want_s.numpy()) < RED_TOL
    # rows wider than the channels that take part (row strides of the C ABI): 64 of the inputs / outputs only
    sub = ops.spatial_wgrad_tile(to_gpu(x), to_gpu(dy), to_gpu(a), cin=64, cout=64)
    want_sub = want.reshape(3, cin, cout)[:, :64, :64].reshape(1, 192, 64)
    assert rel_l2(sub.cpu().numpy(), want_sub.numpy()) < RED_TOL
    par = ops.spatial_wgrad_tile(to_gpu(x), to_gpu(dy), to_gpu(a), conv_param=(3, cin - 3))  # (3, cout, cin_true, 1, 1)
    want_p = want.reshape(3, cin, cout)[:, :cin - 3].permute(0, 2, 1)
    assert tuple(par.shape) == (3, cout, cin - 3, 1, 1)
    assert rel_l2(par.cpu().numpy().reshape(3, cout, cin - 3), want_p.numpy()) < RED_TOL


@pytest.mark.math_modes("bf16x3", "f16x2")
@pytest.mark.parametrize("V,T,cin,ic,B", [(25, 13, 64, 16, 2), (25, 7, 256, 64, 2), (27, 9, 64, 32, 1), (18, 10, 128, 32, 2), (16, 8, 128, 64, 1),
                                          (32, 5, 64, 16, 1), (22, 31, 256, 64, 1), (25, 300, 64, 16, 1), (17, 23, 32, 16, 2), (21, 12, 96, 32, 3)])
def test_embedding_forward_tile_form(V, T, cin, ic, B):
    """The forward of the attention embeddings with the affinity gram on chip (fgcn_emb_fwd_tile.hip; reference agcn.py:104-106): emb = x . Wemb
    + b and the partial grams theta_k^T phi_k against the float64 formulas -- ic = 16 / 32 (all six groups in one workgroup) and 64 (one subset
    per workgroup), 1 .. 8 chunks of input channels, ragged last frame groups, 4 .. 8 frames per tile, other segmentations (tuning key 22), rows
    wider than the channels; zero padding of the 32 x 32 partial matrices; bitwise reproducible; agrees with the 1x1 product + joint_gram."""
    from fusion_gcn_amd import _lib, ops
    if not ops.emb_fwd_tile_available(V, ic, cin):
        pytest.skip("the tile form runs in the split-bf16 math modes")
    x, w, bias = rnd(B, T, V, cin, seed=360), rnd(cin, 6 * ic, seed=361, scale=cin ** -0.5), rnd(6 * ic, seed=362)
    want = x @ w + bias
    e6 = want.reshape(B, T, V, 3, 2, ic)
    want_s = torch.einsum("btvke,btwke->bkvw", e6[..., 0, :], e6[..., 1, :])
    w3 = ops.pack_split3(to_gpu(w.reshape(1, cin, 6 * ic)))
    emb, part = ops.emb_fwd_tile(to_gpu(x), w3, to_gpu(bias), ic=ic)
    assert tuple(emb.shape) == (B, T, V, 6 * ic) and tuple(part.shape[2:]) == (3, 32, 32)
    assert rel_l2(emb.cpu().numpy(), want.numpy()) < FWD_TOL
    got_s = part.double().sum(1)[:, :, :V, :V].cpu()
    assert rel_l2(got_s.numpy(), want_s.numpy()) < RED_TOL
    if V < 32:
        assert float(part[:, :, :, V:, :].abs().max()) == 0.0 and float(part[:, :, :, :, V:].abs().max()) == 0.0     # padding stays zero
    emb2, part2 = ops.emb_fwd_tile(to_gpu(x), w3, to_gpu(bias), ic=ic)
    assert torch.equal(emb, emb2) and torch.equal(part, part2)
    lib = _lib.load()
    segs = {0: part.shape[1]}
    for target in (1, 100000):                  # one long segment per sample / one frame tile per workgroup
        try:
            assert lib.fgcn_set_tuning(22, target) == 0
            e_t, p_t = ops.emb_fwd_tile(to_gpu(x), w3, to_gpu(bias), ic=ic)
        finally:
            assert lib.fgcn_set_tuning(22, 0) == 0
        segs[target] = p_t.shape[1]
        assert torch.equal(e_t, emb)
        assert rel_l2(p_t.double().sum(1)[:, :, :V, :V].cpu().numpy(), want_s.numpy()) < RED_TOL, target
    assert segs[1] == 1 and segs[100000] == (T + 128 // V - 1) // (128 // V), segs
    # the unfused pair on the same data
    emb_old = torch.empty(B, T, V, 6 * ic, device=dev())
    ops.rows_gemm(to_gpu(x), to_gpu(w.reshape(1, cin, 6 * ic)), emb_old, K=cin, N=6 * ic, bias=to_gpu(bias))
    part_old = ops.joint_gram(emb_old, emb_old, [(2 * k * ic, (2 * k + 1) * ic, ic) for k in range(3)])
    assert rel_l2(emb.cpu().numpy(), emb_old.cpu().numpy()) < FWD_TOL
    assert rel_l2(part.sum(1).cpu().numpy(), part_old.sum(1).cpu().numpy()) < RED_TOL
    # ... and through the adjacency softmax
    adj = to_gpu(rnd(3, V, V, seed=363, scale=0.1))
    c_new, a_new = ops.adj_softmax_fwd(part, 1.0 / (ic * T), adj, B)
    c_old, a_old = ops.adj_softmax_fwd(part_old, 1.0 / (ic * T), adj, B)
    assert rel_l2(c_new.cpu().numpy(), c_old.cpu().numpy()) < FWD_TOL and rel_l2(a_new.cpu().numpy(), a_old.cpu().numpy()) < FWD_TOL
    # rows wider than the channels that take part
    wide = torch.zeros(B, T, V, cin + 8, device=dev())
    wide[..., :cin] = to_gpu(x)
    e_w, p_w = ops.emb_fwd_tile(wide, w3, to_gpu(bias), ic=ic, cin=cin)
    assert torch.equal(e_w, emb) and torch.equal(p_w, part)


def emb_backward_reference(emb, ds, x, w, ic):
    """float64 backward of the attention embeddings (agcn.py:104-106): -> demb, demb . W, demb^T . x, column sums of demb."""
    B, T, V, _ = emb.shape
    e = emb.reshape(B, T, V, 3, 2, ic)
    theta, phi = e[..., 0, :], e[..., 1, :]                                  # (B,T,V,3,ic)
    dtheta = torch.einsum("bkvw,btwke->btvke", ds, phi)
    dphi = torch.einsum("bkvw,btvke->btwke", ds, theta)
    demb = torch.stack([dtheta, dphi], dim=4).reshape(B, T, V, 6 * ic)
    return demb, demb @ w, torch.einsum("btvj,btvc->jc", demb, x), demb.sum((0, 1, 2))


EMB_TILE_SHAPES = [(25, 13, 16, 64, 2), (25, 7, 64, 256, 2), (27, 9, 32, 64, 1), (18, 10, 16, 64, 2), (16, 8, 32, 128, 1), (32, 5, 64, 128, 1),
                   (22, 31, 64, 256, 1), (25, 300, 16, 64, 1), (17, 23, 48, 128, 2), (21, 12, 32, 128, 3), (25, 20, 16, 128, 3),
                   (19, 2, 128, 64, 2)]


@pytest.mark.math_modes("bf16x3", "f16x2")
@pytest.mark.parametrize("V,T,ic,cx,B", EMB_TILE_SHAPES)
def test_embedding_backward_tile_form(V, T, ic, cx, B):
    """The backward of the attention embeddings with demb on chip (fgcn_emb_tile.hip; reference agcn.py:104-106): dx (+)= demb . Wemb and
    dWemb = demb^T . x, dbemb = column sums of demb against the float64 formulas -- ic = 16 / 32 / 48 / 64 / 128 (one, two or four
    (subset, side) groups per 64 channels, a partial last channel group), ragged last frame groups, 4 .. 8 frames per tile, one and two
    column tiles, per-sample and shared dS, with and without accumulation, other segmentations of the weight gradient (tuning key 17);
    bitwise reproducible; agrees with the unfused chain joint_mix_vec + row GEMM + row weight gradient."""
    from fusion_gcn_amd import _lib, block, ops
    if not ops.emb_tile_available(V, ic, cx):
        pytest.skip("the tile form runs in the split-bf16 math modes")
    emb, ds = rnd(B, T, V, 6 * ic, seed=350), rnd(B, 3, V, V, seed=351, scale=0.3)
    x, base = rnd(B, T, V, cx, seed=352), rnd(B, T, V, cx, seed=353)
    w = rnd(6 * ic, cx, seed=354, scale=(6 * ic) ** -0.5)                        # Wemb[j][c]
    demb, want_dx, want_w, want_b = emb_backward_reference(emb, ds, x, w, ic)
    w3 = ops.pack_split3(to_gpu(w.reshape(1, 6 * ic, cx)))
    outs = []
    for acc in (False, True):
        dx = to_gpu(base) if acc else torch.full((B, T, V, cx), float("nan"), device=dev())   # without accumulation every element is written
        ops.emb_dx_tile(to_gpu(emb), to_gpu(ds), w3, dx, ic=ic, accumulate=acc)
        assert rel_l2(dx.cpu().numpy(), (want_dx + (base if acc else 0)).numpy()) < FWD_TOL, acc
        outs.append(dx)
    dx2 = to_gpu(base)
    ops.emb_dx_tile(to_gpu(emb), to_gpu(ds), w3, dx2, ic=ic, accumulate=True)
    assert torch.equal(dx2, outs[1])
    gw, gb = ops.emb_wgrad_tile(to_gpu(emb), to_gpu(x), to_gpu(ds), ic=ic)
    assert tuple(gw.shape) == (6 * ic, cx) and tuple(gb.shape) == (6 * ic,)
    assert rel_l2(gw.cpu().numpy(), want_w.numpy()) < RED_TOL
    assert rel_l2(gb.cpu().numpy(), want_b.numpy()) < RED_TOL
    gw2, gb2 = ops.emb_wgrad_tile(to_gpu(emb), to_gpu(x), to_gpu(ds), ic=ic)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    lib = _lib.load()
    slabs = {0: lib.fgcn_emb_wgrad_tile_slabs(B, T, V, ic, cx)}
    for target in (2, 100000):                  # one or two long segments (several samples each) / one (sample, tile) pair per workgroup
        try:
            assert lib.fgcn_set_tuning(17, target) == 0
            slabs[target] = lib.fgcn_emb_wgrad_tile_slabs(B, T, V, ic, cx)
            g2, b2 = ops.emb_wgrad_tile(to_gpu(emb), to_gpu(x), to_gpu(ds), ic=ic)
        finally:
            assert lib.fgcn_set_tuning(17, 0) == 0
        assert rel_l2(g2.cpu().numpy(), want_w.numpy()) < RED_TOL, target
        assert rel_l2(b2.cpu().numpy(), want_b.numpy()) < RED_TOL, target
    assert slabs[2] <= 2 and slabs[2] <= slabs[0] <= slabs[100000], slabs
    for tiles in (0, 2):                        # the widest tiles the channels allow (default) / 64 x 64 tiles
        try:
            assert lib.fgcn_set_tuning(21, tiles) == 0
            gt, bt = ops.emb_wgrad_tile(to_gpu(emb), to_gpu(x), to_gpu(ds), ic=ic)
        finally:
            assert lib.fgcn_set_tuning(21, 0) == 0
        assert rel_l2(gt.cpu().numpy(), want_w.numpy()) < RED_TOL and rel_l2(bt.cpu().numpy(), want_b.numpy()) < RED_TOL, tiles
    # shared dS (one matrix set for every sample)
    _, sh_dx, sh_w, sh_b = emb_backward_reference(emb, ds[:1].expand(B, 3, V, V), x, w, ic)
    dx = torch.empty(B, T, V, cx, device=dev())
    ops.emb_dx_tile(to_gpu(emb), to_gpu(ds[:1]), w3, dx, ic=ic, accumulate=False)
    assert rel_l2(dx.cpu().numpy(), sh_dx.numpy()) < FWD_TOL
    gws, gbs = ops.emb_wgrad_tile(to_gpu(emb), to_gpu(x), to_gpu(ds[:1]), ic=ic)
    assert rel_l2(gws.cpu().numpy(), sh_w.numpy()) < RED_TOL and rel_l2(gbs.cpu().numpy(), sh_b.numpy()) < RED_TOL
    # the unfused chain on the same data
    demb_g = torch.empty(B, T, V, 6 * ic, device=dev())
    sums = block.mix_demb(to_gpu(emb), demb_g, to_gpu(ds), ic)
    dx_old = to_gpu(base)
    ops.rows_gemm(demb_g, to_gpu(w.reshape(1, 6 * ic, cx)), dx_old, K=6 * ic, N=cx, accumulate=True)
    assert rel_l2(outs[1].cpu().numpy(), dx_old.cpu().numpy()) < FWD_TOL
    assert rel_l2(gb.cpu().numpy(), sums.cpu().numpy()) < RED_TOL
    gw_old = ops.rows_wgrad(to_gpu(x), demb_g, K=cx, N=6 * ic)[0]               # (cx, 6 ic)
    assert rel_l2(gw.cpu().numpy(), gw_old.t().cpu().numpy()) < RED_TOL
    # rows wider than the channels that take part (row strides of the C ABI)
    wide_e, wide_x = torch.zeros(B, T, V, 6 * ic + 8, device=dev()), torch.zeros(B, T, V, cx + 4, device=dev())
    wide_e[..., :6 * ic], wide_x[..., :cx] = to_gpu(emb), to_gpu(x)
    wide_dx = torch.zeros(B, T, V, cx + 4, device=dev())
    wide_dx[..., :cx] = to_gpu(base)
    ops.emb_dx_tile(wide_e, to_gpu(ds), w3, wide_dx, ic=ic, accumulate=True, cx=cx)
    assert torch.equal(wide_dx[..., :cx].contiguous(), outs[1]) and float(wide_dx[..., cx:].abs().max()) == 0.0
    gww, gbw = ops.emb_wgrad_tile(wide_e, wide_x, to_gpu(ds), ic=ic, cx=cx)
    assert torch.equal(gww, gw) and torch.equal(gbw, gb)


@pytest.mark.math_modes("bf16x3", "f16x2")
@pytest.mark.parametrize("B,T,V,C", [(3, 20, 25, 64), (2, 13, 18, 128), (1, 40, 25, 256), (2, 9, 27, 64)])
def test_halo_data_gradient_emits_the_batchnorm_backward_sums(B, T, V, C, fgcn_math):
    """fgcn_tconv_halo with bn_a / bn_mask / bn_vec: the data gradient dG and, from its epilogue, sum dP and sum dP * a_hat with
    dP = dG * [G > 0] (backward of G = relu(BN(a) + x), agcn.py:113-115) -- against the float64 formulas and against the
    stand-alone reduction kernel on the same dG; the gradient itself is unchanged by the extra output."""
    from fusion_gcn_amd import ops
    if not ops.tconv_halo_bn_sums():
        pytest.skip("the f32 halo kernel has no BatchNorm-backward epilogue (the block then runs the reduction kernel)")
    kt = 9
    du, wt = rnd(B, T, V, C, seed=120), rnd(kt, C, C, seed=121, scale=(kt * C) ** -0.5)      # packed (kt, o, c) data-gradient form
    a, xres = rnd(B, T, V, C, seed=122), rnd(B, T, V, C, seed=123)
    gamma, beta = rnd(C, seed=124).abs() + 0.5, rnd(C, seed=125)
    mean, var = a.reshape(-1, C).mean(0), a.reshape(-1, C).var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    vec = torch.stack([mean, rstd, gamma * rstd, beta - mean * gamma * rstd])
    g_pre = (a - mean) * rstd * gamma + beta + xres
    _, mask = ops.bn_act(to_gpu(a), to_gpu(vec), to_gpu(xres), None, relu=True, sign_mask=True)
    dg_want = ref_rows_conv(du, wt, ops.conv_dgrad_tmap(kt, 1), T)
    dp = dg_want * (g_pre.float() > 0)
    want = torch.stack([dp.reshape(-1, C).sum(0), (dp * (a - mean) * rstd).reshape(-1, C).sum(0)])
    w4 = ops.pack_conv(to_gpu(wt))
    dg = torch.empty(B, T, V, C, device=dev())
    part = ops.tconv_halo(to_gpu(du), w4, dg, Th=T, taps=kt, tb=-1, tc=4, bn_bwd=(to_gpu(a), mask, to_gpu(vec)))
    assert rel_l2(dg.cpu().numpy(), dg_want.numpy()) < FWD_TOL
    got = part.double().sum(0).cpu()
    assert rel_l2(got.numpy(), want.numpy()) < RED_TOL
    plain = torch.empty_like(dg)
    ops.tconv_halo(to_gpu(du), w4, plain, Th=T, taps=kt, tb=-1, tc=4)
    assert torch.equal(plain, dg)
    _, _, sums = ops.bn_act_bwd(dg, None, to_gpu(a), to_gpu(vec), to_gpu(xres), None, res_mode=1, train=True, sign_mask=mask, need_db=False)
    assert rel_l2(got.numpy(), sums[:2].double().cpu().numpy()) < RED_TOL
    da1, _, s1 = ops.bn_act_bwd(dg, None, to_gpu(a), to_gpu(vec), to_gpu(xres), None, res_mode=1, train=True, sign_mask=mask, need_db=False,
                                partials=part)
    da0, _, _ = ops.bn_act_bwd(dg, None, to_gpu(a), to_gpu(vec), to_gpu(xres), None, res_mode=1, train=True, sign_mask=mask, need_db=False)
    assert rel_l2(da1.cpu().numpy(), da0.cpu().numpy()) < FWD_TOL


@pytest.mark.parametrize("V,T,C,B,n_gated", [(25, 30, 64, 3, 2), (18, 33, 128, 2, 1), (27, 12, 256, 2, 2), (25, 301, 64, 2, 2)])
def test_joint_dagg_adds_the_gated_shortcut_gradients(V, T, C, B, n_gated):
    """dx = sum_k dagg_k . A^_k^T + sum_i e_i * [bit of mask_i]: the ReLU-gated gradients of the identity shortcuts (agcn.py:114,135)
    read from bn_act's one-bit sign images, with and without accumulation; the sign images come from the real bn_act kernel."""
    from fusion_gcn_amd import ops
    x, a = rnd(B, T, V, C, seed=95), rnd(B, 3, V, V, seed=96, scale=0.3)
    dagg, base = rnd(B, T, V, 3 * C, seed=97), rnd(B, T, V, C, seed=98)
    want = torch.einsum("btwkc,bkvw->btvc", dagg.reshape(B, T, V, 3, C), a)
    gated = []
    ident = torch.stack([torch.zeros(C), torch.ones(C), torch.ones(C), torch.zeros(C)]).float().to(dev())   # mean 0, rstd/scale 1, shift 0
    for i in range(n_gated):
        e, pre = rnd(B, T, V, C, seed=100 + i), rnd(B, T, V, C, seed=110 + i)
        out, mask = ops.bn_act(to_gpu(pre), ident, relu=True, sign_mask=True)
        assert mask is not None
        want = want + e * (pre.float() > 0)
        gated.append((to_gpu(e), mask))
    for acc in (False, True):
        dx = to_gpu(base)
        ops.joint_dagg(to_gpu(x), to_gpu(dagg), to_gpu(a), dx, accumulate=acc, gated=gated)
        assert rel_l2(dx.cpu().numpy(), (want + (base if acc else 0)).numpy()) < FWD_TOL


@pytest.mark.parametrize("V,T,ic", [(25, 30, 16), (18, 33, 32), (27, 12, 64), (22, 300, 16)])
def test_joint_gram_and_adjacency_softmax(V, T, ic):
    from fusion_gcn_amd import ops
    B = 3
    emb = rnd(B, T, V, 6 * ic, seed=26)
    e = emb.reshape(B, T, V, 3, 2, ic)
    score = torch.einsum("btvke,btwke->bkvw", e[..., 0, :], e[..., 1, :]) / (ic * T)
    adj_ab = rnd(3, V, V, seed=27, scale=0.2)
    score.requires_grad_(True)
    c_want = torch.softmax(score, dim=-2)
    part = ops.joint_gram(to_gpu(emb), to_gpu(emb), [(2 * k * ic, (2 * k + 1) * ic, ic) for k in range(3)])
    assert part.shape[2:] == (3, 32, 32)
    got_s = part.double().sum(1)[..., :V, :V].cpu() / (ic * T)
    assert rel_l2(got_s.numpy(), score.detach().numpy()) < RED_TOL
    c, a_hat = ops.adj_softmax_fwd(part, 1.0 / (ic * T), to_gpu(adj_ab), B)
    assert rel_l2(c.cpu().numpy(), c_want.detach().numpy()) < 1e-5
    assert rel_l2(a_hat.cpu().numpy(), (c_want.detach() + adj_ab).numpy()) < 1e-5
    np.testing.assert_allclose(c.sum(-2).cpu().numpy(), 1.0, atol=1e-5)
    # backward: partial sums of dA^ -> dA^ and dS through the column softmax
    d_a = rnd(B, 3, V, V, seed=28)
    (ds_want,) = torch.autograd.grad((c_want * d_a).sum(), score)
    fake = torch.zeros(B, 2, 3, 32, 32, dtype=torch.float64)
    fake[:, 0, :, :V, :V] = 0.25 * d_a
    fake[:, 1, :, :V, :V] = 0.75 * d_a
    d_a_hat, d_s = ops.adj_softmax_bwd(to_gpu(fake), 1.0 / (ic * T), c, V)
    assert rel_l2(d_a_hat.cpu().numpy(), d_a.numpy()) < 1e-6
    assert rel_l2(d_s.cpu().numpy(), (ds_want / (ic * T)).numpy()) < 2e-5
    # static adjacency: a_hat is adj_ab, no softmax
    _, a_static = ops.adj_softmax_fwd(None, 1.0, to_gpu(adj_ab), 1, use_softmax=False)
    np.testing.assert_array_equal(a_static[0].cpu().numpy(), adj_ab.float().numpy())


def test_joint_gram_general_operands():
    """dA^_k[v, w] = sum_{t,c} x[t, v, c] dagg[t, w, (k, c)] (two different tensors, channel windows)."""
    from fusion_gcn_amd import ops
    B, T, V, cin = 2, 10, 25, 64
    x, dagg = rnd(B, T, V, cin, seed=29), rnd(B, T, V, 3 * cin, seed=30)
    want = torch.einsum("btvc,btwkc->bkvw", x, dagg.reshape(B, T, V, 3, cin))
    part = ops.joint_gram(to_gpu(x), to_gpu(dagg), [(0, k * cin, cin) for k in range(3)])
    assert rel_l2(part.double().sum(1)[..., :V, :V].cpu().numpy(), want.numpy()) < RED_TOL
    x4, d4 = rnd(B, T, V, 4, seed=31), rnd(B, T, V, 12, seed=32)
    x4[..., 3] = 0
    want = torch.einsum("btvc,btwkc->bkvw", x4, d4.reshape(B, T, V, 3, 4))
    part = ops.joint_gram(to_gpu(x4), to_gpu(d4), [(0, 4 * k, 4) for k in range(3)])
    assert rel_l2(part.double().sum(1)[..., :V, :V].cpu().numpy(), want.numpy()) < RED_TOL


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,res_mode", [(64, 0), (64, 1), (128, 2), (256, 2), (16, 1), (12, 1)])
def test_batchnorm_epilogue_forward_backward(C, res_mode):
    from fusion_gcn_amd import ops
    rows = 3 * 17 * 25
    a = rnd(rows, C, seed=40) * 1.7 + 0.3
    b = rnd(rows, C, seed=41) if res_mode else None
    gam, bet = rnd(C, seed=42).abs() + 0.5, rnd(C, seed=43)
    gam_b, bet_b = rnd(C, seed=44).abs() + 0.5, rnd(C, seed=45)
    a_r = a.clone().requires_grad_(True)
    b_r = b.clone().requires_grad_(True) if b is not None else None

    def bn(t, g_, b_):
        m, v = t.mean(0), t.var(0, unbiased=False)
        return (t - m) / torch.sqrt(v + 1e-5) * g_ + b_
    z = bn(a_r, gam, bet)
    if res_mode == 1:
        z = z + b_r
    elif res_mode == 2:
        z = z + bn(b_r, gam_b, bet_b)
    out_want = torch.relu(z)

    # statistics partials as a producer kernel would emit them (two tiles)
    def partials(t):
        h = rows // 2
        p = torch.stack([torch.stack([t[:h].sum(0), (t[:h] ** 2).sum(0)]), torch.stack([t[h:].sum(0), (t[h:] ** 2).sum(0)])])
        return to_gpu(p)
    rm, rv = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
    vec_a = ops.bn_finalize(partials(a), rows, to_gpu(gam), to_gpu(bet), rm, rv)
    assert rel_l2(vec_a[0].cpu().numpy(), a.mean(0).numpy()) < 1e-6
    assert rel_l2(rm.cpu().numpy(), (0.1 * a.mean(0)).numpy()) < 1e-6
    assert rel_l2(rv.cpu().numpy(), (0.9 + 0.1 * a.var(0, unbiased=True)).numpy()) < 1e-6
    vec_b = ops.bn_finalize(partials(b), rows, to_gpu(gam_b), to_gpu(bet_b)) if res_mode == 2 else None
    ag, bg = to_gpu(a), (to_gpu(b) if b is not None else None)
    out = ops.bn_act(ag, vec_a, bg, vec_b, relu=True)
    assert rel_l2(out.cpu().numpy(), out_want.detach().numpy()) < FWD_TOL
    # the sign-bit image the backward can read instead of `out`: bit e%8 of byte e/8 = [out[e] > 0], same output tensor
    out2, sign = ops.bn_act(ag, vec_a, bg, vec_b, relu=True, sign_mask=True)
    assert torch.equal(out2, out)
    if (rows * C) % 8 == 0:
        want_bits = np.packbits((out.cpu().numpy().reshape(-1) > 0), bitorder="little")
        assert np.array_equal(sign.cpu().numpy(), want_bits)
    else:
        assert sign is None

    dout = rnd(rows, C, seed=46)
    ins = [a_r] + ([b_r] if b_r is not None else [])
    grads = torch.autograd.grad((out_want * dout).sum(), ins + [])
    # use the oracle's ReLU mask so a borderline activation cannot flip the comparison
    out_mask = to_gpu(out_want.detach())
    da, db, sums = ops.bn_act_bwd(to_gpu(dout), out_mask, ag, vec_a, bg, vec_b, res_mode=res_mode, relu=True, train=True)
    assert rel_l2(da.cpu().numpy(), grads[0].numpy()) < 2e-5
    if res_mode:
        assert rel_l2(db.cpu().numpy(), grads[1].numpy()) < 2e-5
    dp = dout * (out_want.detach() > 0)
    a_hat = (a - a.mean(0)) / torch.sqrt(a.var(0, unbiased=False) + 1e-5)
    assert rel_l2(sums[0].cpu().numpy(), dp.sum(0).numpy()) < RED_TOL
    assert rel_l2(sums[1].cpu().numpy(), (dp * a_hat).sum(0).numpy()) < RED_TOL
    if (rows * C) % 8 == 0:          # gate from the bit image: bit-identical to the gate from the tensor
        bits = torch.from_numpy(np.packbits((out_want.detach().numpy().reshape(-1) > 0), bitorder="little")).to(dev())
        da2, db2, sums2 = ops.bn_act_bwd(to_gpu(dout), None, ag, vec_a, bg, vec_b, res_mode=res_mode, relu=True,
                                         train=True, sign_mask=bits)
        assert torch.equal(da2, da) and torch.equal(sums2, sums) and (db is None or torch.equal(db2, db))
    # eval-mode coefficients
    vec_e = ops.bn_eval_coeffs(to_gpu(gam), to_gpu(bet), rm, rv)
    want_scale = gam / torch.sqrt(rv.cpu().double() + 1e-5)
    assert rel_l2(vec_e[2].cpu().numpy(), want_scale.numpy()) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
def test_kernels_are_deterministic():
    """Same inputs, same bits (no atomics anywhere in the path)."""
    from fusion_gcn_amd import ops
    x, w = to_gpu(rnd(4, 50, 25, 64, seed=60)), to_gpu(rnd(9, 64, 64, seed=61, scale=0.05))
    outs = []
    for _ in range(2):
        out = torch.empty(4, 50, 25, 64, device=dev())
        part = ops.rows_gemm(x, w, out, K=64, N=64, tmap=ops.conv_tmap(9, 1), stats=True)
        g = ops.rows_wgrad(x, out, K=64, N=64, tmap=ops.conv_tmap(9, 1))
        outs.append((out.clone(), part.clone(), g.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


# ---- the two ends of the step (fgcn_head.hip): data_bn and CrossEntropy ------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("N,M,T,V,C", [(4, 2, 37, 25, 3), (3, 1, 100, 20, 3), (2, 2, 16, 27, 3), (5, 2, 33, 18, 2), (2, 1, 8, 22, 9)])
@pytest.mark.parametrize("train", [True, False])
def test_data_bn_matches_batchnorm1d(N, M, T, V, C, train):
    """block.data_bn = the reference's permute / view / nn.BatchNorm1d / view / permute (agcn.py:186-188) + the zero pad channel:
    output, running statistics, batch counter, gradients of gamma / beta / x, against torch float64 on the CPU."""
    from fusion_gcn_amd.block import data_bn
    torch.manual_seed(N * 100 + T)
    ch = M * V * C
    ref = torch.nn.BatchNorm1d(ch).double()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5), ref.bias.uniform_(-0.5, 0.5)
        ref.running_mean.uniform_(-0.3, 0.3), ref.running_var.uniform_(0.5, 2.0)
    mine = torch.nn.BatchNorm1d(ch)
    mine.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in ref.state_dict().items()})
    mine = mine.to(dev())
    ref.train(train), mine.train(train)
    x = (torch.randn(N, M, T, V, C, dtype=torch.float64) * 2 + 0.3)
    xr = x.clone().requires_grad_(True)
    h = ref(xr.permute(0, 1, 3, 4, 2).contiguous().view(N, ch, T)).view(N, M, V, C, T).permute(0, 1, 4, 2, 3).reshape(N * M, T, V, C)
    xm = x.float().to(dev()).requires_grad_(True)
    got = data_bn(xm, mine)
    Cp = (C + 3) // 4 * 4
    assert got.shape == (N * M, T, V, Cp)
    assert rel_l2(got[..., :C].detach().cpu().numpy(), h.detach().numpy()) < 2e-6
    assert float(got[..., C:].abs().max()) == 0.0 if Cp > C else True
    probe = torch.randn(N * M, T, V, Cp, dtype=torch.float64)
    (h * probe[..., :C]).sum().backward()
    (got * probe.float().to(dev())).sum().backward()
    assert rel_l2(mine.weight.grad.cpu().numpy(), ref.weight.grad.numpy()) < 1e-5
    assert rel_l2(mine.bias.grad.cpu().numpy(), ref.bias.grad.numpy()) < 1e-5
    assert rel_l2(xm.grad.cpu().numpy(), xr.grad.numpy()) < 2e-5
    for k in ("running_mean", "running_var"):
        assert rel_l2(getattr(mine, k).cpu().numpy(), getattr(ref, k).numpy()) < 2e-6, k
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked)
    # fixed-order sums: a second run gives the same bits
    mine.zero_grad()
    xm2 = x.float().to(dev()).requires_grad_(True)
    got2 = data_bn(xm2, mine) if not train else None
    if got2 is not None:
        assert torch.equal(got2, got)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,classes", [(64, 60), (2, 27), (7, 35), (130, 60), (1, 5), (16, 200)])
def test_cross_entropy_matches_torch(rows, classes):
    """fusion_gcn_amd.loss.CrossEntropyLoss = nn.CrossEntropyLoss() (mean; ignore_index rows do not count), on logits that are a
    column window of a padded matrix (what the classifier hands over), forward and gradient, bitwise repeatable."""
    import torch.nn.functional as F
    from fusion_gcn_amd.loss import CrossEntropyLoss
    torch.manual_seed(rows + classes)
    npad = (classes + 3) // 4 * 4
    z = torch.randn(rows, npad, dtype=torch.float64) * 3
    y = torch.randint(0, classes, (rows,))
    if rows > 4:
        y[1] = -100                                              # torch's ignore_index
    zr = z[:, :classes].clone().requires_grad_(True)
    want = F.cross_entropy(zr, y)
    (want * 1.7).backward()
    base = z.float().to(dev()).requires_grad_(True)
    loss_fn = CrossEntropyLoss()
    got = loss_fn(base[:, :classes], y.to(dev()))
    (got * 1.7).backward()
    assert abs(float(got) - float(want)) < 2e-6 * max(1.0, abs(float(want)))
    assert rel_l2(base.grad[:, :classes].cpu().numpy(), zr.grad.numpy()) < 2e-6
    assert float(base.grad[:, classes:].abs().max()) == 0.0 if npad > classes else True
    again = loss_fn(base.detach()[:, :classes], y.to(dev()))
    assert torch.equal(again, got.detach())
    # a label outside [0, classes) that is not ignore_index (torch: a device assert) fails loudly: NaN loss and gradients
    bad = y.clone()
    bad[0] = classes + 3
    z2 = z.float().to(dev()).requires_grad_(True)
    lb = loss_fn(z2[:, :classes], bad.to(dev()))
    lb.backward()
    assert torch.isnan(lb) and torch.isnan(z2.grad[0, :classes]).all()


@pytest.mark.math_modes("f16x2")
@pytest.mark.parametrize("B,T,V,K,N,kt,stride", [(2, 40, 25, 64, 64, 9, 1), (3, 33, 25, 128, 128, 9, 1), (2, 21, 25, 128, 128, 9, 2),
                                                 (2, 16, 27, 256, 256, 9, 1), (2, 30, 25, 128, 192, 1, 1), (2, 24, 22, 256, 384, 1, 1)])
def test_weight_gradient_from_two_way_f16_splits(B, T, V, K, N, kt, stride):
    """Math mode f16x2: the all-taps / 1x1 weight gradient with its operands scaled by the maxima that the data-path kernels of the
    same tensors recorded (tconv_halo / pw_gemm ``amax_out``) -- against float64, at the f32 tolerance; the recorded maxima are
    the tensors' true maxima; operands spanning 2^-20 .. 2^20 around unit scale keep the accuracy (power-of-two scaling is exact)."""
    from fusion_gcn_amd import ops
    if ops.get_math_mode() != "f16x2":
        pytest.skip("f16x2 products only")
    for scale_a, scale_g in ((1.0, 1.0), (2.0 ** -20, 2.0 ** 17)):
        Tg = (T - 1) // stride + 1
        a, g = rnd(B, T, V, K, seed=300) * scale_a, rnd(B, Tg, V, N, seed=301) * scale_g
        ag, gg = to_gpu(a), to_gpu(g)
        slots = torch.zeros(2, device=dev(), dtype=torch.int32)
        if kt > 1:
            # what the block does: forward conv stages a (= G), data gradient stages g (= dU)
            w = rnd(kt, K, N, seed=302, scale=(kt * K) ** -0.5)
            wf, wb = ops.pack_conv(to_gpu(w)), ops.pack_conv(to_gpu(w.permute(0, 2, 1)))
            if stride == 1:
                ops.tconv_halo(ag, wf, torch.empty(B, Tg, V, N, device=dev()), Th=T, taps=kt, tb=1, tc=-4, amax_out=slots[0:1])
                ops.tconv_halo(gg, wb, torch.empty(B, T, V, K, device=dev()), Th=T, taps=kt, tb=-1, tc=4, amax_out=slots[1:2])
            else:
                slots[0] = torch.tensor(float(ag.abs().max())).view(torch.int32)      # (any writer of the true maxima will do)
                slots[1] = torch.tensor(float(gg.abs().max())).view(torch.int32)
            got = ops.tconv_wgrad(ag, gg, taps=kt, stride=stride, amax=(slots[0:1], slots[1:2]))
            tm = ops.conv_tmap(kt, stride)
        else:
            wa = ops.pack_conv(to_gpu(rnd(1, K, 64, seed=303, scale=K ** -0.5)))
            wg_ = ops.pack_conv(to_gpu(rnd(1, N, 64, seed=304, scale=N ** -0.5)))
            ops.pw_gemm(ag, wa, torch.empty(B, T, V, 64, device=dev()), amax_out=slots[0:1])
            ops.pw_gemm(gg, wg_, torch.empty(B, Tg, V, 64, device=dev()), amax_out=slots[1:2])
            got = ops.rows_wgrad(ag, gg, K=K, N=N, amax=(slots[0:1], slots[1:2]))
            tm = ops.TMAP_POINTWISE
        rec = slots.view(torch.float32).cpu()
        assert float(rec[0]) == float(ag.abs().max()) and float(rec[1]) == float(gg.abs().max())
        want = torch.zeros(kt, K, N, dtype=torch.float64)
        taps, ta, tb, tc, td = tm
        for to in range(Tg):
            for j in range(taps):
                num = to * ta + j * tb + tc
                if num < 0 or num % td or num // td >= T:
                    continue
                want[j] += torch.einsum("bvk,bvn->kn", a[:, num // td], g[:, to])
        assert rel_l2(got.cpu().numpy().reshape(kt, K, N), want.numpy()) < RED_TOL, (scale_a, scale_g)


# ---- inference forms of the two north-star kernels (include/fgcn.h: fgcn_tconv_halo_bn_relu, fgcn_spatial_fwd_tile_bn_relu) ------------------
def _eval_vec(C, seed):
    """(4, C) vector of an eval-mode BatchNorm as fgcn_bn_eval_coeffs lays it out: mean, rstd, scale, shift"""
    mean, var = rnd(C, seed=seed), rnd(C, seed=seed + 1).abs() + 0.5
    gamma, beta = rnd(C, seed=seed + 2), rnd(C, seed=seed + 3)
    rstd = (var + 1e-5).rsqrt()
    return torch.stack([mean, rstd, gamma * rstd, beta - mean * gamma * rstd]).float()


@pytest.mark.math_modes("bf16x3")
@pytest.mark.parametrize("B,T,V,C,kt,res", [(3, 37, 25, 64, 9, "identity"), (2, 21, 25, 128, 9, "identity"), (2, 9, 27, 256, 9, "conv"),
                                            (1, 50, 22, 64, 9, "none"), (2, 40, 32, 128, 5, "identity"), (130, 5, 18, 64, 3, "conv")])
def test_temporal_conv_with_batchnorm_shortcut_and_relu_in_one_kernel(B, T, V, C, kt, res):
    """North-star kernel 2 as the north star states it, inference form (fgcn_tconv_halo_bn_relu): out = relu(BN(conv(g) + bias) +
    [x | BN_r(r)]) against float64 and against the two-kernel form (fgcn_tconv_halo + fgcn_bn_act with the same eval-mode vectors),
    bitwise repeatable; all three shortcut kinds, 64 / 128 / 256 channels (every tile form the launcher picks), ragged tiles."""
    from fusion_gcn_amd import ops
    g, w = rnd(B, T, V, C, seed=90), rnd(kt, C, C, seed=91, scale=(kt * C) ** -0.5)
    bias, vec = rnd(C, seed=92), _eval_vec(C, 93)
    r, rvec = rnd(B, T, V, C, seed=97), (_eval_vec(C, 98) if res == "conv" else None)
    pad = (kt - 1) // 2
    u = torch.zeros(B, T, V, C, dtype=torch.float64)
    for j in range(kt):
        for t in range(T):
            ti = t + j - pad
            if 0 <= ti < T:
                u[:, t] += g[:, ti].double() @ w[j].double()
    z = (u + bias.double()) * vec[2].double() + vec[3].double()
    if res == "identity":
        z = z + r.double()
    elif res == "conv":
        z = z + r.double() * rvec[2].double() + rvec[3].double()
    want = z.clamp_min(0)
    w4 = ops.pack_split3(to_gpu(w))
    rg = None if res == "none" else to_gpu(r)
    rv = None if rvec is None else to_gpu(rvec)
    out = torch.empty(B, T, V, C, device=dev())
    ops.tconv_halo_bn_relu(to_gpu(g), w4, out, taps=kt, tb=1, tc=-pad, vec=to_gpu(vec), bias=to_gpu(bias), res=rg, res_vec=rv)
    assert rel_l2(out.cpu().numpy(), want.numpy()) < FWD_TOL
    again = torch.empty_like(out)
    ops.tconv_halo_bn_relu(to_gpu(g), w4, again, taps=kt, tb=1, tc=-pad, vec=to_gpu(vec), bias=to_gpu(bias), res=rg, res_vec=rv)
    assert torch.equal(out, again)
    ug = torch.empty(B, T, V, C, device=dev())
    ops.tconv_halo(to_gpu(g), w4, ug, Th=T, taps=kt, tb=1, tc=-pad, bias=to_gpu(bias))
    two = ops.bn_act(ug, to_gpu(vec), rg, rv, relu=True)
    assert rel_l2(out.cpu().numpy(), two.cpu().numpy()) < 2e-6


@pytest.mark.math_modes("bf16x3")
@pytest.mark.parametrize("V,T,cin,cout,B,down", [(25, 13, 64, 64, 2, False), (25, 7, 128, 256, 2, True), (27, 9, 64, 128, 1, True),
                                                  (18, 10, 128, 128, 2, False), (22, 31, 256, 256, 1, False), (32, 5, 64, 64, 3, False)])
def test_spatial_stage_with_batchnorm_shortcut_and_relu_in_one_kernel(V, T, cin, cout, B, down):
    """North-star kernel 1, inference form (fgcn_spatial_fwd_tile_bn_relu): g = relu(BN(sum_k (x . A^_k) . Wd_k + bias) + [x | BN_d(d)])
    against float64 and against fgcn_spatial_fwd_tile + fgcn_bn_act; identity shortcut (cin == cout) and the down branch."""
    from fusion_gcn_amd import ops
    x, a = rnd(B, T, V, cin, seed=300), rnd(B, 3, V, V, seed=301, scale=0.3)
    wd, bias = rnd(3, cin, cout, seed=302, scale=(3 * cin) ** -0.5), rnd(cout, seed=303)
    vec = _eval_vec(cout, 304)
    d, dvec = (rnd(B, T, V, cout, seed=308), _eval_vec(cout, 309)) if down else (None, None)
    y = torch.einsum("btwkc,kco->btwo", torch.einsum("btvc,bkvw->btwkc", x.double(), a.double()), wd.double()) + bias.double()
    z = y * vec[2].double() + vec[3].double()
    z = z + (d.double() * dvec[2].double() + dvec[3].double() if down else x.double())
    want = z.clamp_min(0)
    w3 = ops.pack_split3(to_gpu(wd.reshape(1, 3 * cin, cout)))
    res = to_gpu(d) if down else to_gpu(x)
    rv = to_gpu(dvec) if down else None
    g = ops.spatial_fwd_tile_bn_relu(to_gpu(x), to_gpu(a), w3, to_gpu(bias), to_gpu(vec), Cin=cin, Cout=cout, res=res, res_vec=rv)
    assert rel_l2(g.cpu().numpy(), want.numpy()) < FWD_TOL
    assert torch.equal(g, ops.spatial_fwd_tile_bn_relu(to_gpu(x), to_gpu(a), w3, to_gpu(bias), to_gpu(vec), Cin=cin, Cout=cout, res=res, res_vec=rv))
    yg, _ = ops.spatial_fwd_tile(to_gpu(x), to_gpu(a), w3, to_gpu(bias), Cin=cin, Cout=cout, stats=False)
    two = ops.bn_act(yg, to_gpu(vec), res, rv, relu=True)
    assert rel_l2(g.cpu().numpy(), two.cpu().numpy()) < 2e-6
