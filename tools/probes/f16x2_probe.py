"""Probe: f32-class products from TWO-way f16 splits (three MFMAs per product group instead of bf16x3's six) on the persistent 1x1
row GEMM (fgcn_probe_pw_gemm_f16x2, fgcn_pw.hip) -- speed and accuracy against the bf16x3 form of the same kernel and float64.
Operands are scaled by powers of two so that their largest magnitude sits near 2^14 (f16 keeps 11 bits down to 2^-14; below that the
low part goes subnormal and the representation error becomes an absolute 2^-25 of the scaled value).  Not a math mode of the library."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from fusion_gcn_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
fn = lib.fgcn_probe_pw_gemm_f16x2
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_void_p]


def pow2_scale(t, target=14):
    return 2.0 ** (target - math.frexp(float(t.abs().max()))[1])


def pack2h(w, scale):
    """(K, N) f32 -> [2][K/8][N][8] f16 halves of w * scale"""
    ws = (w.double() * scale)
    h = ws.half()
    lo = (ws - h.double()).half()
    K, N = w.shape
    return torch.stack([t.view(K // 8, 8, N).permute(0, 2, 1).contiguous() for t in (h, lo)]).contiguous()


def timeit(f, reps=20):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def run(rows, K, N, spread=0.0, label=""):
    g = torch.Generator().manual_seed(K + N)
    x = torch.randn(rows, K, generator=g, dtype=torch.float64)
    if spread:          # per-element magnitudes spread over `spread` binades (log-uniform), as gradients behind ReLU gates are
        x = x * torch.exp2(-spread * torch.rand(rows, K, generator=g, dtype=torch.float64))
    w = torch.randn(K, N, generator=g, dtype=torch.float64) * K ** -0.5
    xg, wg = x.float().to(dev), w.float().to(dev)
    want = xg.double() @ wg.double()
    sa, sw = pow2_scale(xg), pow2_scale(wg)
    w2 = pack2h(wg, sw)
    out = torch.empty(rows, N, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def f16x2():
        _lib.check(fn(xg.data_ptr(), out.data_ptr(), w2.data_ptr(), None, rows, K, N, K, N, sa, sw, stream), "probe")
    f16x2()
    e_h = float((out.double() - want).norm() / want.norm())
    m_h = float((out.double() - want).abs().max() / want.abs().max())
    ms_h = timeit(f16x2)
    with ops.math_mode("bf16x3"):
        w3 = ops.pack_split3(wg.view(1, K, N).contiguous())
        o3 = torch.empty(rows, 1, 1, N, device=dev)
        ops.pw_gemm(xg.view(rows, 1, 1, K), w3, o3)
        e_3 = float((o3.view(rows, N).double() - want).norm() / want.norm())
        ms_3 = timeit(lambda: ops.pw_gemm(xg.view(rows, 1, 1, K), w3, o3))
    e_f = float(((xg @ wg).double() - want).norm() / want.norm())          # torch's f32 GEMM (hipBLASLt) on the same data
    fl = 2.0 * rows * K * N
    print(f"{label:14s} rows {rows} K {K:3d} N {N:3d}: f16x2 {ms_h:.3f} ms {fl / ms_h / 1e9:6.1f} TF/s rel-L2 {e_h:.2e} max/amax {m_h:.1e} | "
          f"bf16x3 {ms_3:.3f} ms {fl / ms_3 / 1e9:6.1f} TF/s rel-L2 {e_3:.2e} | torch f32 GEMM rel-L2 {e_f:.2e} | speed-up {ms_3 / ms_h:.2f}x")


for rows, K, N in ((480000, 128, 192), (480000, 128, 384), (240000, 256, 384), (240000, 256, 768), (240000, 384, 256), (960000, 64, 192)):
    run(rows, K, N, label="normal data")
run(240000, 256, 384, spread=12.0, label="12 binades")
run(240000, 256, 384, spread=24.0, label="24 binades")
run(240000, 256, 384, spread=40.0, label="40 binades")
