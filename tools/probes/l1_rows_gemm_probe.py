"""The first block's 1x1 convolutions (4 padded input channels) on the exact-f32 row GEMM, one by one (rocprofv3 --kernel-trace --stats):
emb 4 -> 96, down 4 -> 64 (+ BatchNorm sums), and the data gradients 96 -> 4 (accumulating) / 64 -> 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fusion_gcn_amd import ops
dev = torch.device("cuda:0")
B, T, V = 128, 300, 25
with ops.math_mode("bf16x3"):
    x = torch.randn(B, T, V, 4, device=dev)
    for (K, N, stats, acc, tag) in ((4, 96, False, False, "emb"), (4, 64, True, False, "down"), (96, 4, False, False, "demb->dx"), (64, 4, False, True, "dd->dx acc")):
        a = torch.randn(B, T, V, K, device=dev)
        w = torch.randn(1, K, N, device=dev)
        out = torch.zeros(B, T, V, N, device=dev)
        bias = torch.randn(N, device=dev)
        for _ in range(3):
            ops.rows_gemm(a, w, out, K=K, N=N, bias=bias if not acc else None, stats=stats, accumulate=acc)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            ops.rows_gemm(a, w, out, K=K, N=N, bias=bias if not acc else None, stats=stats, accumulate=acc)
        e.record(); torch.cuda.synchronize()
        byts = 4.0 * B * T * V * (K + N * (2 if acc else 1))
        print(f"{tag:14s} K={K:3d} N={N:3d}: {s.elapsed_time(e) / 10 * 1e3:7.1f} us  {byts / (s.elapsed_time(e) / 10 * 1e-3) / 1e12:5.2f} TB/s")
