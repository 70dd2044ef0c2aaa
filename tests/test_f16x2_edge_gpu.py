"""Hardening of math mode "f16x2" (include/fgcn.h FGCN_PRODUCTS_F16X2): inside bf16x3, the temporal / 1x1
convolutions and their weight gradients form every float32 product from TWO-way f16 splits of both operands -- three MFMAs instead of
six -- with every operand block scaled by an exact power of two so that its largest magnitude sits in [2^14, 2^15):

    x 2^s = h + l,  h = f16(x 2^s),  l = f16(x 2^s - h):   |x 2^s - h - l| <= 2^-24 |x 2^s|  while l is a normal f16 (|x 2^s| >= 2^-2),
                                                            an absolute 2^-25 (in scaled units) below that;   l.l is dropped.

Scaling blocks: activations of the forward / data-gradient kernels per staged (128-row tile + halo) x (32 | 64 channel) chunk, inside
the kernel, the accumulators following the scale exactly; weights per packed form; both operands of a weight gradient per whole
tensor (the maxima the data-path kernels recorded).  So the error model per operand element is max(2^-24 |x|, 2^-40 block maximum):

  * what is asserted at the f32 contract (error <= 2x the exact-f32-MFMA error on the same data, like bf16x3): O(1) data,
    cancellation-heavy sums, magnitudes from 2^-120 to 2^100, activation channel chunks (in either order) spanning 32 binades against
    weight rows spanning 16, neighbouring samples 12 binades apart;
  * what is NOT f32-class, stated and bounded here instead of hidden: operands whose elements span many binades INSIDE one scaling
    block while the other operand is scaled inversely (bf16x3's "80 binades per contraction index" case): an element 2^-d below its
    block's maximum keeps 2^-(40 - d) relative accuracy -- full f32 accuracy down to d = 16, nothing below d = 40.  Activations
    behind a BatchNorm and gradients of one layer do not look like that; the mode's parity tests (kernels, blocks, model, full-size
    model, end-to-end gradients: the same tolerances as f32) are what pins its use on this path.
Reference semantics: torch_src/models/mmargcn/agcn.py:41-42 (Conv2d 9x1), :71-73,77 (1x1)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RATIO = 2.0
MODES = ("f32", "bf16x3", "f16x2")


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def f32(x):
    return x.float().double()


def gpu(x):
    return x.float().to(dev()).contiguous()


def pow2(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.pow(2.0, torch.randint(lo, hi + 1, shape, generator=g).double())


def conv_ref(x, w, kt):
    B, T, V, K = x.shape
    pad = (kt - 1) // 2
    out = torch.zeros(B, T, V, w.shape[2], dtype=torch.float64)
    mag = torch.zeros_like(out)
    for j in range(kt):
        lo, hi = max(0, pad - j), min(T, T + pad - j)
        out[:, lo:hi] += x[:, lo + j - pad:hi + j - pad] @ w[j]
        mag[:, lo:hi] += x[:, lo + j - pad:hi + j - pad].abs() @ w[j].abs()
    return out, mag


def all_modes(fn):
    from fusion_gcn_amd import ops
    out = {}
    for mode in MODES:
        with ops.math_mode(mode):
            out[mode] = fn().double().cpu()
    return out


def errors(got, want, mag):
    return {m: float(((g - want).abs() / mag.clamp_min(1e-300)).max()) for m, g in got.items()}


def check(tag, got, want, mag, ratio=RATIO):
    err = errors(got, want, mag)
    print(f"[{tag}] max |err| / sum|terms|: f32 MFMA {err['f32']:.2e}, bf16x3 {err['bf16x3']:.2e}, f16x2 {err['f16x2']:.2e}")
    assert np.isfinite(err["f16x2"]) and err["f16x2"] <= ratio * err["f32"] + 4 * 2.0 ** -24, (tag, err)
    return err


def conv(x, w, kt):
    from fusion_gcn_amd import ops
    B, T, V, K = x.shape

    def run():
        out = torch.empty(B, T, V, w.shape[2], device=dev())
        ops.tconv_halo(gpu(x), ops.pack_conv(gpu(w)), out, Th=T, taps=kt, tb=1, tc=-((kt - 1) // 2))
        return out
    return all_modes(run)


def pointwise(x, w):
    from fusion_gcn_amd import ops
    rows = x.numel() // x.shape[-1]

    def run():
        if not ops.pw_gemm_available():         # math mode f32: the exact-f32 row GEMM is this product's kernel
            out = torch.empty(*x.shape[:-1], w.shape[2], device=dev())
            ops.rows_gemm(gpu(x), gpu(w), out, K=w.shape[1], N=w.shape[2])
            return out
        out = torch.empty(rows, 1, 1, w.shape[2], device=dev())
        ops.pw_gemm(gpu(x).view(rows, 1, 1, -1), ops.pack_conv(gpu(w)), out)
        return out.view(*x.shape[:-1], w.shape[2])
    return all_modes(run)


@pytest.mark.parametrize("C,kt", [(64, 9), (128, 9), (256, 9), (256, 1)])
def test_unit_scale_and_cancellation(C, kt):
    B, T, V = 2, 12, 25
    x, w = f32(rnd(B, T, V, C, seed=3)), f32(rnd(kt, C, C, seed=4, scale=(kt * C) ** -0.5))
    want, mag = conv_ref(x, w, kt)
    check(f"unit scale conv {C}ch {kt}tap", conv(x, w, kt), want, mag)
    x2 = rnd(B, T, V, C, seed=5)
    x2[..., 1::2] = x2[..., 0::2] * (1 + 1e-3 * rnd(B, T, V, C // 2, seed=6))
    w2 = rnd(kt, C, C, seed=7, scale=(kt * C) ** -0.5)
    w2[:, 1::2] = -w2[:, 0::2]
    x2, w2 = f32(x2), f32(w2)
    want, mag = conv_ref(x2, w2, kt)
    assert float((want.abs() / mag).median()) < 5e-3            # the sums really cancel
    check(f"cancellation conv {C}ch {kt}tap", conv(x2, w2, kt), want, mag)


@pytest.mark.parametrize("log2_x,log2_w", [(-100, 100), (100, -100), (-60, -40), (40, 30), (-120, 0)])
def test_global_magnitudes_from_2_to_minus_100_to_2_to_100(log2_x, log2_w):
    """Power-of-two block scales make the split independent of the operands' overall magnitude (f16 alone would over- or underflow)."""
    C, kt, B, T, V = 64, 9, 2, 8, 25
    x = f32(rnd(B, T, V, C, seed=8) * 2.0 ** log2_x)
    w = f32(rnd(kt, C, C, seed=9, scale=(kt * C) ** -0.5) * 2.0 ** log2_w)
    want, mag = conv_ref(x, w, kt)
    check(f"conv x 2^{log2_x} w 2^{log2_w}", conv(x, w, kt), want, mag)
    w1 = f32(rnd(1, 128, 192, seed=10, scale=128 ** -0.5) * 2.0 ** log2_w)
    x1 = f32(rnd(2, 40, 25, 128, seed=11) * 2.0 ** log2_x)
    want1, mag1 = conv_ref(x1, w1, 1)
    check(f"1x1 x 2^{log2_x} w 2^{log2_w}", pointwise(x1, w1), want1, mag1)


@pytest.mark.parametrize("order", ["rising", "falling", "random"])
def test_blocks_spanning_40_binades(order):
    """32- / 64-channel chunks of the activations on scales 2^-20 .. 2^20: each staged block is split at its own scale and the
    accumulators follow, in either direction.  The weights are ONE block per packed form, so their rows are only scaled inversely by
    half as many binades (2^-8 .. 2^8: elements within 2^16 of the form's maximum keep f32 accuracy) -- the chunks' products then span
    2^-12 .. 2^12.  Samples sit on scales 2^-6 .. 2^6: a 128-row tile may straddle two samples, and the rows of the smaller one are
    split at the larger one's scale (an output row keeps f32 accuracy while its inputs are within 2^16 of its tile's maximum: the
    stated limit, next test)."""
    B, T, V, kt = 4, 10, 25, 9
    for C, chunk in ((128, 32), (256, 32)):
        nchunk = C // chunk
        e = torch.linspace(-20, 20, nchunk).round()
        e = 2 * (e / 2.5).round()                                # even exponents: the weights take half of each, inverted
        if order == "falling":
            e = e.flip(0)
        elif order == "random":
            e = e[torch.randperm(nchunk, generator=torch.Generator().manual_seed(3))]
        g_c = torch.pow(2.0, e).repeat_interleave(chunk)
        s_b = pow2((B, 1, 1, 1), -6, 6, seed=12)
        x = f32(rnd(B, T, V, C, seed=13) * g_c * s_b)
        w = f32(rnd(kt, C, C, seed=14, scale=(kt * C) ** -0.5) / g_c.sqrt()[None, :, None])
        want, mag = conv_ref(x, w, kt)
        check(f"{order} chunk scales, conv {C}ch", conv(x, w, kt), want, mag)
    K, N, chunk = 256, 384, 64
    e = 2 * (torch.linspace(-20, 20, K // chunk) / 2.5).round()
    e = e.flip(0) if order == "falling" else e
    g_c = torch.pow(2.0, e).repeat_interleave(chunk)
    x = f32(rnd(3, 50, 25, K, seed=15) * g_c * pow2((3, 1, 1, 1), -6, 6, seed=16))
    w = f32(rnd(1, K, N, seed=17, scale=K ** -0.5) / g_c.sqrt()[None, :, None])
    want, mag = conv_ref(x, w, 1)
    check(f"{order} chunk scales, 1x1 {K}->{N}", pointwise(x, w), want, mag)


@pytest.mark.parametrize("spread", [8, 16, 24, 32])
def test_dynamic_range_inside_a_block_is_the_stated_limit(spread):
    """Channels of ONE chunk on scales 2^0 .. 2^-spread with the weight rows scaled inversely: the small channels' products matter as
    much as the large ones', but their elements sit `spread` binades below the block maximum and keep 2^-(40 - spread) relative
    accuracy.  Asserted: f32-class up to 16 binades, and never worse than the model max(2^-24, 2^-(38 - spread)) -- reported either way."""
    C, kt, B, T, V = 64, 9, 2, 8, 25
    g_c = torch.pow(2.0, -torch.linspace(0, spread, 32).round()).repeat(C // 32)
    x = f32(rnd(B, T, V, C, seed=18) * g_c)
    w = f32(rnd(kt, C, C, seed=19, scale=(kt * C) ** -0.5) / g_c[None, :, None])
    want, mag = conv_ref(x, w, kt)
    got = conv(x, w, kt)
    err = errors(got, want, mag)
    print(f"[{spread} binades inside a chunk] max |err| / sum|terms|: f32 MFMA {err['f32']:.2e}, bf16x3 {err['bf16x3']:.2e}, "
          f"f16x2 {err['f16x2']:.2e}")
    if spread <= 16:
        assert err["f16x2"] <= RATIO * err["f32"] + 4 * 2.0 ** -24, err
    assert np.isfinite(err["f16x2"]) and err["f16x2"] <= max(2.0 ** -22, 2.0 ** -(38 - spread)), err


def test_all_zero_chunks_and_weights():
    from fusion_gcn_amd import ops
    C, kt, B, T, V = 128, 9, 2, 8, 25
    x = rnd(B, T, V, C, seed=20)
    x[..., 32:96] = 0                                             # two all-zero chunks in the middle
    x[1] = 0                                                      # an all-zero sample
    w = rnd(kt, C, C, seed=21, scale=(kt * C) ** -0.5)
    x, w = f32(x), f32(w)
    want, mag = conv_ref(x, w, kt)
    got = conv(x, w, kt)
    assert float(got["f16x2"][1].abs().max()) == 0.0
    check("zero chunks", {m: g[:1] for m, g in got.items()}, want[:1], mag[:1])
    with ops.math_mode("f16x2"):
        out = torch.full((B, T, V, C), 7.0, device=dev())
        ops.tconv_halo(gpu(x), ops.pack_conv(gpu(torch.zeros_like(w))), out, Th=T, taps=kt, tb=1, tc=-4)
        assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("C,kt", [(64, 9), (128, 9), (128, 1)])
def test_weight_gradient_at_tensor_scale(C, kt):
    """dW = sum_rows a . g with whole-tensor scales (the maxima recorded by the data-path kernels): O(1) data on any overall
    magnitude and a cancelling half at the f32 contract; rows scaled over `d` binades against inverse scales on g degrade as stated
    (tensor-level block) -- reported for d = 8, 20."""
    from fusion_gcn_amd import ops
    B, T, V = 2, 24, 25
    pad = (kt - 1) // 2

    def run_case(a, g, tag, assert_f32_class=True):
        want = torch.zeros(kt, C, C, dtype=torch.float64)
        mag = torch.zeros_like(want)
        for j in range(kt):
            lo, hi = max(0, pad - j), min(T, T + pad - j)
            want[j] = torch.einsum("btvk,btvn->kn", a[:, lo + j - pad:hi + j - pad], g[:, lo:hi])
            mag[j] = torch.einsum("btvk,btvn->kn", a[:, lo + j - pad:hi + j - pad].abs(), g[:, lo:hi].abs())

        def run():
            ag, gg = gpu(a), gpu(g)
            slots = torch.stack([ag.abs().max(), gg.abs().max()]).view(torch.int32)       # what tconv_halo / pw_gemm record
            amax = (slots[0:1], slots[1:2]) if ops.get_math_mode() == "f16x2" else None
            if kt > 1:
                return ops.tconv_wgrad(ag, gg, taps=kt, stride=1, amax=amax).reshape(kt, C, C)
            return ops.rows_wgrad(ag, gg, K=C, N=C, amax=amax).reshape(kt, C, C)
        got = all_modes(run)
        if assert_f32_class:
            return check(tag, got, want, mag)
        err = errors(got, want, mag)
        print(f"[{tag}] max |err| / sum|terms|: f32 MFMA {err['f32']:.2e}, bf16x3 {err['bf16x3']:.2e}, f16x2 {err['f16x2']:.2e}")
        return err

    a, g = rnd(B, T, V, C, seed=22), rnd(B, T, V, C, seed=23)
    run_case(f32(a * 2.0 ** -70), f32(g * 2.0 ** 50), f"wgrad {C}ch {kt}tap, a 2^-70 g 2^50")
    g2 = g.clone()
    g2[:, :, 1::2] = -g2[:, :, 0:-1:2] * (1 + 1e-3)
    a2 = a.clone()
    a2[:, :, 1::2] = a2[:, :, 0:-1:2]
    run_case(f32(a2), f32(g2), f"wgrad {C}ch {kt}tap, cancelling joints")
    for d in (8, 20):
        e = pow2((B, T, V, 1), -d, 0, seed=24)
        err = run_case(f32(a * e), f32(g / e), f"wgrad {C}ch {kt}tap, rows over {d} binades, inverse on g", assert_f32_class=(d <= 8))
        assert err["f16x2"] <= max(2.0 ** -22, 2.0 ** -(38 - d)), err      # an element d binades below its tensor's maximum: 2^-(40 - d)


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 256)])
@pytest.mark.parametrize("log2_x,log2_a,log2_w", [(0, 0, 0), (-90, 10, 60), (70, -20, -40)])
def test_fused_spatial_forward_scales(cin, cout, log2_x, log2_a, log2_w):
    """y = sum_k (x . A^_k) . Wd_k in the f16x2 form of the fused kernel: x is scaled per (frame pair, channel tile), A^ per sample, the
    aggregation it forms in registers per (tile, subset), the weights per form, and the step-2 accumulators follow the product of the
    four -- any overall magnitude of the three operands, and channel tiles of x on different scales, at the f32 contract."""
    from fusion_gcn_amd import ops
    B, T, V = 2, 6, 25
    tile_scale = torch.pow(2.0, torch.tensor([0.0, -9.0, 7.0, -3.0, 5.0, -6.0, 2.0, 8.0])[:cin // 32]).repeat_interleave(32)
    x = f32(rnd(B, T, V, cin, seed=15) * tile_scale * 2.0 ** log2_x)
    a = f32(rnd(B, 3, V, V, seed=16, scale=0.3) * 2.0 ** log2_a)
    wd = f32(rnd(3, cin, cout, seed=17, scale=(3 * cin) ** -0.5) * 2.0 ** log2_w)
    agg = torch.einsum("btvc,bkvw->btwkc", x, a)
    want = torch.einsum("btwkc,kco->btwo", agg, wd)
    mag = torch.einsum("btwkc,kco->btwo", torch.einsum("btvc,bkvw->btwkc", x.abs(), a.abs()), wd.abs())

    def run():
        y, _ = ops.spatial_fwd(gpu(x), gpu(a), ops.pack_spatial(gpu(wd.reshape(3 * cin, cout)), cin), None, Cin=cin, Cout=cout,
                               stats=False)
        return y
    check(f"spatial_fwd {cin}->{cout} x 2^{log2_x} A 2^{log2_a} W 2^{log2_w}", all_modes(run), want, mag)
