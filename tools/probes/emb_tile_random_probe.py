#!/usr/bin/env python3
"""Random-shape stress of the two round-5 embedding-backward tile kernels (fgcn_emb_dx_tile, fgcn_emb_wgrad_tile) against the float64
formulas of agcn.py:104-106's backward (run on the GPU box: PYTHONPATH=. python3 tools/probes/emb_tile_random_probe.py [cases] [seed]).
Shapes: V in 16..32, T in 1..40, B in 1..5, ic in 16s up to 128, input channels in 64s up to 256, shared or per-sample dS, random row
strides, random workgroup targets / tile widths (tuning keys 17, 21), math modes bf16x3 and f16x2.  Prints the worst relative errors;
exits non-zero above the suite's tolerances."""
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
    lib = _lib.load()
    worst = {"dx": 0.0, "dW": 0.0, "db": 0.0}
    done = 0
    for i in range(10 * cases):
        if done == cases:
            break
        V, T, B = rng.randint(16, 32), rng.randint(1, 40), rng.randint(1, 5)
        ic, cx = 16 * rng.randint(1, 8), 64 * rng.randint(1, 4)
        ops.set_math_mode(rng.choice(("bf16x3", "f16x2")))
        if not ops.emb_tile_available(V, ic, cx):
            continue
        done += 1
        shared = rng.random() < 0.3
        pad_e, pad_x = 4 * rng.randint(0, 3), 4 * rng.randint(0, 3)
        g = torch.Generator(device="cuda").manual_seed(i)
        emb = torch.randn(B, T, V, 6 * ic + pad_e, device="cuda", generator=g)
        x = torch.randn(B, T, V, cx + pad_x, device="cuda", generator=g)
        ds = torch.randn(1 if shared else B, 3, V, V, device="cuda", generator=g) * 0.3
        w = torch.randn(6 * ic, cx, device="cuda", generator=g) * (6 * ic) ** -0.5
        base = torch.randn(B, T, V, cx + pad_x, device="cuda", generator=g)
        e6 = emb[..., :6 * ic].double().reshape(B, T, V, 3, 2, ic)
        dsb = ds.double().expand(B, 3, V, V)
        dth = torch.einsum("bkvw,btwke->btvke", dsb, e6[..., 1, :])
        dph = torch.einsum("bkvw,btvke->btwke", dsb, e6[..., 0, :])
        demb = torch.stack([dth, dph], dim=4).reshape(B, T, V, 6 * ic)
        want_dx = base[..., :cx].double() + demb @ w.double()
        want_w = torch.einsum("btvj,btvc->jc", demb, x[..., :cx].double())
        want_b = demb.sum((0, 1, 2))
        assert lib.fgcn_set_tuning(17, rng.choice((0, 1, 7, 64, 100000))) == 0
        assert lib.fgcn_set_tuning(21, rng.choice((0, 2))) == 0
        dx = base.clone()
        ops.emb_dx_tile(emb, ds, ops.pack_split3(w.reshape(1, 6 * ic, cx).contiguous()), dx, ic=ic, accumulate=True, cx=cx)
        gw, gb = ops.emb_wgrad_tile(emb, x, ds, ic=ic, cx=cx)
        e = (rel(dx[..., :cx], want_dx), rel(gw, want_w), rel(gb, want_b))
        worst = {k: max(worst[k], v) for k, v in zip(("dx", "dW", "db"), e)}
        untouched = bool(torch.equal(dx[..., cx:], base[..., cx:]))
        ok = e[0] < FWD_TOL and e[1] < RED_TOL and e[2] < RED_TOL and untouched and bool(torch.isfinite(gw).all())
        print(f"{done:3d} {ops.get_math_mode():6s} V={V} T={T} B={B} ic={ic} cx={cx} shared={int(shared)} ld+{pad_e}/{pad_x}  dx {e[0]:.2e} dW {e[1]:.2e} "
              f"db {e[2]:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        if not ok:
            sys.exit(1)
    lib.fgcn_set_tuning(17, 0), lib.fgcn_set_tuning(21, 0)
    print("worst", worst)


if __name__ == "__main__":
    main()
