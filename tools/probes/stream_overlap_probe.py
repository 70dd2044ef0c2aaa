"""Overlap probe: an MFMA-bound weight-gradient kernel beside an HBM-bound BN backward on two streams."""
import sys, torch
sys.path.insert(0, ".")
from fusion_gcn_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def t_ms(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for prio in (0, 1, -1):
  side = torch.cuda.Stream(priority=prio)
  print("side priority", prio, "->", side.priority)
  for C, T in ((64, 300), (128, 150), (256, 75)):
    B, V = 128, 25
    g = torch.randn(B, T, V, C, device=dev); du = torch.randn(B, T, V, C, device=dev)
    y = torch.randn(B, T, V, C, device=dev); x = torch.randn(B, T, V, C, device=dev)
    part = torch.randn(64, 2, C, device=dev).abs()
    vec = ops.bn_finalize(part.contiguous(), B * T * V, torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)) if False else None
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    # BN vector via the public path
    yy, st = ops.rows_gemm(x, torch.eye(C, device=dev).view(1, C, C).contiguous(), torch.empty_like(x), K=C, N=C, stats=True), None
    dx = torch.empty_like(x)
    import inspect
    def mfma():
        ops.tconv_wgrad(g, du, taps=9, stride=1, conv_param=(1, C))
    vec_y = None
    def hbm():
        ops.bn_act(y, VEC, x, None, relu=True)
        ops.bn_act(y, VEC, x, None, relu=True)
        ops.bn_act(y, VEC, x, None, relu=True)
    # make a BN vector (mean, rstd, scale, shift) by hand
    VEC = torch.cat([torch.zeros(C), torch.ones(C), torch.ones(C), torch.zeros(C)]).to(dev)
    main = torch.cuda.current_stream()
    def both():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            mfma()
        hbm()
        main.wait_stream(side)
    a, b, ab = t_ms(mfma), t_ms(hbm), t_ms(both)
    print(f"C={C}: wgrad {a:.3f} ms, 3x bn_act {b:.3f} ms, concurrent {ab:.3f} ms (sum {a+b:.3f}, max {max(a,b):.3f})")
