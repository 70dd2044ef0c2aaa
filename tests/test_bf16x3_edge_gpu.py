"""Hardening of the benchmark's arithmetic (math mode "bf16x3", include/fgcn.h FGCN_MATH_BF16X3): every
f32 product from exact three-way bfloat16 splits of both operands (split_bf16_pair, fgcn_common.hpp), six partial products.

The claim under test is "same accuracy as the f32 MFMA path" -- so each case runs the SAME kernel on the SAME data in both math
modes and compares both with the float64 result: the bf16x3 error must be <= 2x the exact-f32-MFMA error (+ a floor of a few
f32 ulps of the sum of |terms|, below which both are rounding noise).  Data is chosen to stress the split, not O(1) noise:

  binades       operands spanning 80 binades per contraction index (x[..., k] * 2^g_k, w[k, :] * 2^-g_k, g_k in [-40, 40]) and
                per row (x[row] * 2^e_row): every high / middle / low term is formed at many different exponents;
  cancellation  pairs of contraction indices that cancel to 1e-3 of their magnitude: the result is what the low terms carry;
  tiny          magnitudes down to where the LOW term of an operand becomes a bf16 subnormal (|x| < 2^-109): reported, and
                asserted down to 2^-100 (activations and weights of the model are O(1e-4 .. 1e2)).
Kernels: the halo-tile 9x1 temporal convolution (dominant kernel), its one-tap 1x1 form, the all-tap weight gradient, the fused
spatial forward.  Reference semantics: torch_src/models/mmargcn/agcn.py:41-42 (Conv2d), :109-111 (x . A^ and conv_d)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RATIO = 2.0


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def f32(x):
    """what the kernels are handed: the float32 rounding of the data, as float64 for the reference product"""
    return x.float().double()


def gpu(x):
    return x.float().to(dev()).contiguous()


def pow2(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.pow(2.0, torch.randint(lo, hi + 1, shape, generator=g).double())


def conv_ref(x, w, kt):
    """x (B,T,V,K), w (kt,K,N), stride 1, 'same' padding; also returns sum of |terms| (the scale rounding errors live on)"""
    B, T, V, K = x.shape
    pad = (kt - 1) // 2
    out = torch.zeros(B, T, V, w.shape[2], dtype=torch.float64)
    mag = torch.zeros_like(out)
    for j in range(kt):
        lo, hi = max(0, pad - j), min(T, T + pad - j)
        out[:, lo:hi] += x[:, lo + j - pad:hi + j - pad] @ w[j]
        mag[:, lo:hi] += x[:, lo + j - pad:hi + j - pad].abs() @ w[j].abs()
    return out, mag


def both_modes(fn):
    from fusion_gcn_amd import ops
    out = {}
    for mode in ("f32", "bf16x3"):
        with ops.math_mode(mode):
            out[mode] = fn().double().cpu()
    return out


def check(tag, got, want, mag, ratio=RATIO):
    """errors relative to the sum of |terms| per output (max over outputs); bf16x3 <= ratio * f32 + 4 ulp"""
    err = {m: float(((g - want).abs() / mag.clamp_min(1e-300)).max()) for m, g in got.items()}
    print(f"[{tag}] max |err| / sum|terms|: f32 MFMA {err['f32']:.2e}, bf16x3 {err['bf16x3']:.2e}")
    assert np.isfinite(err["bf16x3"]) and err["bf16x3"] <= ratio * err["f32"] + 4 * 2.0 ** -24, (tag, err)
    return err


def halo_conv(x, w, kt):
    from fusion_gcn_amd import ops
    B, T, V, K = x.shape

    def run():
        out = torch.empty(B, T, V, w.shape[2], device=dev())
        ops.tconv_halo(gpu(x), ops.pack_conv(gpu(w)), out, Th=T, taps=kt, tb=1, tc=-((kt - 1) // 2))
        return out
    return both_modes(run)


@pytest.mark.parametrize("C,kt", [(64, 9), (128, 9), (256, 1)])
def test_halo_conv_operands_spanning_80_binades(C, kt):
    B, T, V = 2, 12, 25
    g_k = pow2((C,), -40, 40, seed=1)
    e_row = pow2((B, T, V, 1), -20, 20, seed=2)
    x = f32(rnd(B, T, V, C, seed=3) * g_k * e_row)
    w = f32(rnd(kt, C, C, seed=4, scale=(kt * C) ** -0.5) / g_k[None, :, None])
    want, mag = conv_ref(x, w, kt)
    check(f"binades conv {C}ch {kt}tap", halo_conv(x, w, kt), want, mag)


@pytest.mark.parametrize("C,kt", [(64, 9), (256, 1)])
def test_halo_conv_cancellation_heavy_sums(C, kt):
    B, T, V = 2, 10, 25
    x = rnd(B, T, V, C, seed=5)
    x[..., 1::2] = x[..., 0::2] * (1 + 1e-3 * rnd(B, T, V, C // 2, seed=6))
    w = rnd(kt, C, C, seed=7, scale=(kt * C) ** -0.5)
    w[:, 1::2] = -w[:, 0::2]
    x, w = f32(x), f32(w)
    want, mag = conv_ref(x, w, kt)
    assert float((want.abs() / mag).median()) < 5e-3            # the sums really cancel
    check(f"cancellation conv {C}ch {kt}tap", halo_conv(x, w, kt), want, mag)


@pytest.mark.parametrize("log2_mag", [-60, -100, -112, -120])
def test_halo_conv_tiny_magnitudes(log2_mag):
    """x ~ 2^log2_mag against w ~ 2^-log2_mag (products O(1)): the low split term of x is ~2^(log2_mag - 17).  Down to 2^-100 it is
    a normal bf16 and the result keeps f32 accuracy (asserted); below 2^-109 it becomes a bf16 subnormal -- reported only."""
    C, kt, B, T, V = 64, 9, 2, 8, 25
    x = f32(rnd(B, T, V, C, seed=8) * 2.0 ** log2_mag)
    w = f32(rnd(kt, C, C, seed=9, scale=(kt * C) ** -0.5) * 2.0 ** -log2_mag)
    want, mag = conv_ref(x, w, kt)
    got = halo_conv(x, w, kt)
    if log2_mag >= -100:
        check(f"tiny conv 2^{log2_mag}", got, want, mag)
    else:
        err = {m: float(((g - want).abs() / mag).max()) for m, g in got.items()}
        print(f"[tiny conv 2^{log2_mag}] (low term bf16-subnormal) f32 MFMA {err['f32']:.2e}, bf16x3 {err['bf16x3']:.2e}")
        assert np.isfinite(err["bf16x3"]) and err["bf16x3"] < 2.0 ** -15      # never worse than dropping the low term


@pytest.mark.parametrize("C,kt", [(64, 9), (128, 9)])
def test_weight_gradient_binades_and_cancellation(C, kt):
    """dW[j, k, n] = sum_rows a[row + j - pad, k] g[row, n]: rows scaled over 40 binades (a * 2^e, g * 2^-e) and a cancelling half."""
    from fusion_gcn_amd import ops
    B, T, V = 2, 24, 25
    e = pow2((B, T, V, 1), -20, 20, seed=10)
    a = rnd(B, T, V, C, seed=11)
    g = rnd(B, T, V, C, seed=12)
    g[:, :, 1::2] = -g[:, :, 0:-1:2] * (1 + 1e-3)              # neighbouring joints nearly cancel where a is similar
    a[:, :, 1::2] = a[:, :, 0:-1:2]
    # the same power of two on a row of a and (inverted) on the same row of g only lines up for the centre tap; the other taps
    # see ratios of neighbouring frames' scales: up to 2^40 between terms of one sum
    a, g = f32(a * e), f32(g / e)
    pad = (kt - 1) // 2
    want = torch.zeros(kt, C, C, dtype=torch.float64)
    mag = torch.zeros_like(want)
    for j in range(kt):
        lo, hi = max(0, pad - j), min(T, T + pad - j)
        want[j] = torch.einsum("btvk,btvn->kn", a[:, lo + j - pad:hi + j - pad], g[:, lo:hi])
        mag[j] = torch.einsum("btvk,btvn->kn", a[:, lo + j - pad:hi + j - pad].abs(), g[:, lo:hi].abs())
    got = both_modes(lambda: ops.tconv_wgrad(gpu(a), gpu(g), taps=kt, stride=1))
    check(f"wgrad {C}ch", got, want, mag)


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 256)])
def test_fused_spatial_forward_binades(cin, cout):
    """y = sum_k (x . A^_k) . Wd_k with x's channels and Wd's rows on inverse power-of-two scales, and joint rows of x scaled
    against the columns of A^ (both contraction steps see operands spanning many binades)."""
    from fusion_gcn_amd import ops
    B, T, V = 2, 6, 25
    g_c = pow2((cin,), -30, 30, seed=13)
    s_v = pow2((V,), -10, 10, seed=14)
    x = f32(rnd(B, T, V, cin, seed=15) * g_c * s_v[:, None])
    a = f32(rnd(B, 3, V, V, seed=16, scale=0.3) / s_v[:, None])
    wd = f32(rnd(3, cin, cout, seed=17, scale=(3 * cin) ** -0.5) / g_c[None, :, None])
    agg = torch.einsum("btvc,bkvw->btwkc", x, a)
    want = torch.einsum("btwkc,kco->btwo", agg, wd)
    mag = torch.einsum("btwkc,kco->btwo", torch.einsum("btvc,bkvw->btwkc", x.abs(), a.abs()), wd.abs())

    def run():
        y, _ = ops.spatial_fwd(gpu(x), gpu(a), ops.pack_spatial(gpu(wd.reshape(3 * cin, cout)), cin), None, Cin=cin, Cout=cout,
                               stats=False)
        return y
    check(f"spatial_fwd {cin}->{cout}", both_modes(run), want, mag)
