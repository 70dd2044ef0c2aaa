#!/usr/bin/env python3
"""Random-shape stress of the two round-4 tile kernels against float64 einsums (run on the GPU box: PYTHONPATH=. python3
tools/probes/tile_kernels_random_probe.py [cases] [seed]).  Shapes: V in 16..32, T in 1..40, B in 1..5, channels in 64s up to 256, shared or
per-sample adjacency, random workgroup targets (tuning keys 15 / 16).  Prints the worst relative errors; exits non-zero above the suite's
tolerances."""
import random
import sys

import torch

from fusion_gcn_amd import _lib, ops

FWD_TOL, RED_TOL = 3e-6, 2e-5


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ops.set_math_mode("bf16x3")
    lib = _lib.load()
    worst = {"dW": 0.0, "dx": 0.0, "dA": 0.0}
    for i in range(cases):
        V, T, B = rng.randint(16, 32), rng.randint(1, 40), rng.randint(1, 5)
        cin, cout = 64 * rng.randint(1, 4), 64 * rng.randint(1, 4)
        shared = rng.random() < 0.3
        g = torch.Generator(device="cuda").manual_seed(i)
        x = torch.randn(B, T, V, cin, device="cuda", generator=g)
        dy = torch.randn(B, T, V, cout, device="cuda", generator=g)
        a = torch.randn(1 if shared else B, 3, V, V, device="cuda", generator=g) * 0.3
        ab = a.expand(B, 3, V, V)
        assert lib.fgcn_set_tuning(15, rng.choice((0, 1, 7, 64, 100000))) == 0
        assert lib.fgcn_set_tuning(16, rng.choice((0, 1, 7, 64, 100000))) == 0
        want_w = torch.einsum("btvc,bkvw,btwo->kco", x.double(), ab.double(), dy.double()).reshape(1, 3 * cin, cout)
        got_w = ops.spatial_wgrad_tile(x, dy, a)
        wd = torch.randn(3, cout, cin, device="cuda", generator=g) * cout ** -0.5
        dagg = torch.einsum("btwo,koc->btwkc", dy.double(), wd.double())
        want_dx = torch.einsum("btwkc,bkvw->btvc", dagg, ab.double())
        want_g = torch.einsum("btvc,btwkc->bkvw", x.double(), dagg)
        w3 = ops.pack_split3(wd.permute(1, 0, 2).reshape(1, cout, 3 * cin).contiguous())
        base = torch.randn(B, T, V, cin, device="cuda", generator=g)
        dx = base.clone()
        part = ops.spatial_bwd_tile(dy, x, a, w3, dx, accumulate=True)
        e = (rel(got_w, want_w), rel(dx, want_dx + base.double()), rel(part.double().sum(1)[:, :, :V, :V], want_g))
        worst = {k: max(worst[k], v) for k, v in zip(("dW", "dx", "dA"), e)}
        ok = e[0] < RED_TOL and e[1] < FWD_TOL and e[2] < RED_TOL and bool(torch.isfinite(got_w).all())
        print(f"{i:3d} V={V} T={T} B={B} {cin}->{cout} shared={int(shared)}  dW {e[0]:.2e} dx {e[1]:.2e} dA {e[2]:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        if not ok:
            sys.exit(1)
    lib.fgcn_set_tuning(15, 0), lib.fgcn_set_tuning(16, 0)
    print("worst", worst)


if __name__ == "__main__":
    main()
