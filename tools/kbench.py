#!/usr/bin/env python3
"""Kernel microbenchmark at the headline model's shapes (B = 128 samples unless --b): times every libfgcn kernel
family with HIP events on torch's current stream and prints ms, TFLOP/s (algorithmic) and GB/s (algorithmic).
Used to A/B kernel variants (fgcn_set_tuning) in one process on one device.

    python tools/kbench.py [--b 128] [--reps 10] [--only gemm,wgrad,spatial,joint,elem] [--tune 5=1]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fusion_gcn_amd import _lib, block, ops  # noqa: E402

V = 25
DEV = torch.device("cuda:0")
TUNE = {}          # --tune pairs (sections that flip a key for an A/B line restore the requested value)


def timeit(fn, reps):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def report(name, ms, flops=0.0, byts=0.0):
    print(f"{name:58s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF/s  {byts / ms / 1e6:8.0f} GB/s", flush=True)


def rnd(*shape):
    return torch.randn(*shape, device=DEV)


def bench_gemm(B, reps):
    lib = _lib.load()
    cases = [("tconv fwd", 300, 300, 64, 64, 9, 1), ("tconv fwd", 150, 150, 128, 128, 9, 1),
             ("tconv fwd", 75, 75, 256, 256, 9, 1), ("tconv fwd s2", 300, 150, 128, 128, 9, 2),
             ("emb 1x1", 300, 300, 64, 96, 1, 1), ("emb 1x1", 150, 150, 128, 192, 1, 1),
             ("emb 1x1", 75, 75, 256, 384, 1, 1), ("dagg 1x1", 300, 300, 64, 192, 1, 1),
             ("dagg 1x1", 150, 150, 128, 384, 1, 1), ("dagg 1x1", 75, 75, 256, 768, 1, 1),
             ("demb 1x1", 300, 300, 96, 64, 1, 1), ("demb 1x1", 150, 150, 192, 128, 1, 1),
             ("demb 1x1", 75, 75, 384, 256, 1, 1), ("down 1x1", 300, 300, 64, 128, 1, 1)]
    for k0, k1 in ((1, 0),):
        lib.fgcn_set_tuning(0, k0)
        lib.fgcn_set_tuning(1, k1)
        print(f"-- rows_gemm, tuning small={k0} wide={k1}")
        for name, tin, tout, K, N, kt, s in cases:
            x, w, y = rnd(B, tin, V, K), rnd(kt, K, N) * (kt * K) ** -0.5, torch.empty(B, tout, V, N, device=DEV)
            tm = ops.conv_tmap(kt, s)
            ms = timeit(lambda: ops.rows_gemm(x, w, y, K=K, N=N, tmap=tm, stats=True), reps)
            rows = B * tout * V
            report(f"rows_gemm {name} T{tin}->{tout} K{kt}x{K} N{N}", ms, 2.0 * rows * kt * K * N,
                   4.0 * (B * tin * V * K + rows * N))
            if kt == 1 and K % 32 == 0 and ops.get_math_mode() == "bf16x3":
                w3 = ops.pack_split3(w)
                ms = timeit(lambda: ops.tconv_halo(x, w3, y, Th=tout, taps=1, tb=1, tc=0, stats=True), reps)
                report(f"  same on the split-bf16 halo kernel (taps = 1)", ms, 2.0 * rows * K * N, 4.0 * (rows * K + rows * N))
    lib.fgcn_set_tuning(0, 1)
    lib.fgcn_set_tuning(1, 0)
    # data gradient of the strided conv (half of the taps are empty for every output frame)
    du, wt, dg = rnd(B, 150, V, 128), rnd(9, 128, 128) * 0.03, torch.empty(B, 300, V, 128, device=DEV)
    ms = timeit(lambda: ops.rows_gemm(du, wt, dg, K=128, N=128, tmap=ops.conv_dgrad_tmap(9, 2)), reps)
    report("rows_gemm tconv dgrad s2 T150->300 K9x128 N128", ms, 2.0 * B * 150 * V * 9 * 128 * 128, 4.0 * B * 450 * V * 128)


def bench_pw(B, reps):
    """1x1 convolutions of the block at their headline shapes: exact-f32 row GEMM / split-bf16 halo kernel with one tap / the
    persistent split-bf16 row GEMM (ops.pw_gemm), one tile per workgroup (tuning key 8 = 1) and persistent."""
    lib = _lib.load()
    if not ops.pw_gemm_available():
        print("-- pw: split-bf16 modes only")
        return
    cases = [("emb", 300, 64, 96), ("emb", 150, 128, 192), ("emb", 75, 256, 384), ("d_t", 300, 64, 192), ("d_t", 150, 128, 384),
             ("d_t", 75, 256, 768), ("emb_t", 300, 96, 64), ("emb_t", 150, 192, 128), ("emb_t", 75, 384, 256), ("down", 300, 64, 128),
             ("down_t", 150, 128, 64), ("down", 150, 128, 256), ("down_t", 75, 256, 128)]
    for name, T, K, N in cases:
        x, w, y = rnd(B, T, V, K), rnd(1, K, N) * K ** -0.5, torch.empty(B, T, V, N, device=DEV)
        rows = B * T * V
        fl, by = 2.0 * rows * K * N, 4.0 * rows * (K + N)
        ms = timeit(lambda: ops.rows_gemm(x, w, y, K=K, N=N, stats=True), reps)
        report(f"{name:7s} K{K:3d} N{N:3d} T{T}: f32 row GEMM", ms, fl, by)
        w3 = ops.pack_conv(w)
        if K % 64 == 0:
            ms = timeit(lambda: ops.tconv_halo(x, w3, y, Th=T, taps=1, tb=1, tc=0, stats=True), reps)
            report("            halo kernel, one tap", ms, fl, by)
        lib.fgcn_set_tuning(8, 1)
        ms = timeit(lambda: ops.pw_gemm(x, w3, y, stats=True), reps)
        report("            pw_gemm, one tile per workgroup", ms, fl, by)
        lib.fgcn_set_tuning(8, 0)
        ms = timeit(lambda: ops.pw_gemm(x, w3, y, stats=True), reps)
        report("            pw_gemm, persistent", ms, fl, by)
        ms = timeit(lambda: ops.pw_gemm(x, w3, y, accumulate=True), reps)
        report("            pw_gemm, persistent, accumulate", ms, fl, by + 4.0 * rows * N)


def bench_tconv(B, reps):
    """Halo-tile temporal conv: forward / data gradient, stride 1 and the parity-split stride 2; tuning key 4 picks the
    register budget (2 or 3 workgroups per CU)."""
    lib = _lib.load()
    x3 = ops.get_math_mode() in ops.X3_MODES
    for three in ((0,) if x3 else (0, 1)):
        lib.fgcn_set_tuning(4, three)
        print("-- conv_halo" + ("" if x3 else f", workgroups per CU hint = {3 - three}"))
        for T, c, s in ((300, 64, 1), (150, 128, 1), (75, 256, 1), (300, 128, 2), (150, 256, 2)):
            Tp = (T - 1) // s + 1
            wt = rnd(9, c, c) * (9 * c) ** -0.5
            W = {"t": wt, "t_t": wt.permute(0, 2, 1).contiguous()}
            if s == 1:
                W["t4"], W["t_t4"] = ops.pack_conv(wt), ops.pack_conv(wt.permute(0, 2, 1).contiguous())
            else:
                for par, tag in ((0, "e"), (1, "o")):
                    W[f"t4_{tag}"] = ops.pack_conv(wt[par::2].contiguous())
                    W[f"t_t4_{tag}"] = ops.pack_conv(wt.permute(0, 2, 1)[par::2].contiguous())
            g, u, bias = rnd(B, T, V, c), torch.empty(B, Tp, V, c, device=DEV), rnd(c)
            du, dg = rnd(B, Tp, V, c), torch.empty(B, T, V, c, device=DEV)
            fl = 2.0 * B * Tp * V * 9 * c * c
            ms = timeit(lambda: block.temporal_fwd(g, u, W, bias, 9, s, stats=True), reps)
            report(f"tconv_halo fwd   T{T} s{s} C{c}", ms, fl, 4.0 * B * V * c * (T + Tp))
            ms = timeit(lambda: block.temporal_dgrad(du, dg, W, 9, s), reps)
            report(f"tconv_halo dgrad T{T} s{s} C{c}", ms, fl, 4.0 * B * V * c * (T + Tp))
    lib.fgcn_set_tuning(4, 0)


def bench_wgrad(B, reps):
    cases = [("tconv", 300, 300, 64, 64, 9, 1), ("tconv", 150, 150, 128, 128, 9, 1), ("tconv", 75, 75, 256, 256, 9, 1),
             ("tconv s2", 300, 150, 128, 128, 9, 2), ("tconv s2", 150, 75, 256, 256, 9, 2),
             ("conv_d", 300, 300, 192, 64, 1, 1), ("conv_d", 150, 150, 384, 128, 1, 1), ("conv_d", 75, 75, 768, 256, 1, 1),
             ("emb", 300, 300, 64, 96, 1, 1), ("emb", 150, 150, 128, 192, 1, 1), ("emb", 75, 75, 256, 384, 1, 1),
             ("down", 300, 300, 64, 128, 1, 1), ("down", 150, 150, 128, 256, 1, 1), ("res s2", 300, 150, 64, 128, 1, 2)]
    for name, ta, tg, K, N, kt, s in cases:
        a, g = rnd(B, ta, V, K), rnd(B, tg, V, N)
        tm = ops.conv_tmap(kt, s)
        ms = timeit(lambda: ops.rows_wgrad(a, g, K=K, N=N, tmap=tm, wide=False), reps)
        report(f"rows_wgrad {name} T{ta} K{kt}x{K} N{N}", ms, 2.0 * B * tg * V * kt * K * N, 4.0 * B * V * (ta * K + tg * N))
        if kt == 1:
            ms = timeit(lambda: ops.rows_wgrad(a, g, K=K, N=N, tmap=tm, wide=True), reps)
            report(f"pw_wgrad {name} T{ta} K{K} N{N} (channel chunks)", ms, 2.0 * B * tg * V * K * N,
                   4.0 * B * V * (ta * K + tg * N))
            if ops.get_math_mode() == "bf16x3":
                _lib.load().fgcn_set_tuning(6, TUNE.get(6, 0) | 1)
                ms = timeit(lambda: ops.rows_wgrad(a, g, K=K, N=N, tmap=tm, wide=True), reps)
                _lib.load().fgcn_set_tuning(6, TUNE.get(6, 0))
                report("  same, fragments split as they are read (256-thread kernel)", ms, 2.0 * B * tg * V * K * N,
                       4.0 * B * V * (ta * K + tg * N))
        if kt > 1:
            ms = timeit(lambda: ops.tconv_wgrad(a, g, taps=kt, stride=s, all_taps=True), reps)
            report(f"tconv_wgrad {name} T{ta} K{kt}x{K} N{N} (all taps)", ms, 2.0 * B * tg * V * kt * K * N,
                   4.0 * B * V * (ta * K + tg * N))


def bench_spatial(B, reps):
    for T, cin, cout in ((300, 4, 64), (300, 64, 64), (300, 64, 128), (150, 128, 128), (150, 128, 256), (75, 256, 256)):
        x, a = rnd(B, T, V, cin), rnd(B, 3, V, V) * 0.2
        wd, bias = ops.pack_spatial(rnd(3 * cin, cout) * (3 * cin) ** -0.5, cin), rnd(cout)
        ms = timeit(lambda: ops.spatial_fwd(x, a, wd, bias, Cin=cin, Cout=cout, stats=True), reps)
        rows = B * T * V
        report(f"spatial_fwd T{T} {cin}->{cout}", ms, rows * (6.0 * V * cin + 6.0 * cin * cout), 4.0 * rows * (cin + cout))
        if ops.spatial_fwd_tile_available(V, cin, cout):
            w3 = ops.pack_split3(rnd(1, 3 * cin, cout) * (3 * cin) ** -0.5)
            ms = timeit(lambda: ops.spatial_fwd_tile(x, a, w3, bias, Cin=cin, Cout=cout, stats=True), reps)
            report("  tile form (128 / V frames per workgroup, aggregation image in LDS)", ms, rows * (6.0 * V * cin + 6.0 * cin * cout),
                   4.0 * rows * (cin + cout))
        if ops.get_math_mode() == "bf16x3" and cin % 32 == 0:
            _lib.load().fgcn_set_tuning(7, 1)
            ms = timeit(lambda: ops.spatial_fwd(x, a, wd, bias, Cin=cin, Cout=cout, stats=True), reps)
            _lib.load().fgcn_set_tuning(7, 0)
            report("  same, one frame per wave, f32 aggregation (older form)", ms, rows * (6.0 * V * cin + 6.0 * cin * cout),
                   4.0 * rows * (cin + cout))


def bench_spatial_wgrad(B, reps):
    """conv_d weight gradient: fused agg recompute vs joint_mix_vec + row weight gradient."""
    for T, cin, cout in ((300, 64, 64), (300, 64, 128), (150, 128, 128), (150, 128, 256), (75, 256, 256)):
        x, a, dy = rnd(B, T, V, cin), rnd(B, 3, V, V) * 0.2, rnd(B, T, V, cout)
        agg = torch.empty(B, T, V, 3 * cin, device=DEV)
        rows = B * T * V
        fl = rows * (6.0 * V * cin + 6.0 * cin * cout)
        if ops.spatial_wgrad_tile_available(V, cin, cout):
            ms = timeit(lambda: ops.spatial_wgrad_tile(x, dy, a), reps)
            report(f"spatial_wgrad_tile       T{T} {cin}->{cout}", ms, fl, 4.0 * rows * (cin + cout))
            # the same with operands that are NOT resident in the Infinity Cache (what the step sees): a ring of input sets > 256 MB
            nset = max(2, int(1.2e9 / (4.0 * rows * (cin + cout))))
            ring = [(rnd(B, T, V, cin), rnd(B, T, V, cout)) for _ in range(nset)]
            it = [0]

            def cold():
                xs, dys = ring[it[0] % nset]
                it[0] += 1
                ops.spatial_wgrad_tile(xs, dys, a)
            ms = timeit(cold, max(reps, 2 * nset))
            report(f"  ... operands from HBM (ring of {nset} sets)", ms, fl, 4.0 * rows * (cin + cout))
            del ring
        ms = timeit(lambda: ops.spatial_wgrad(x, dy, a), reps)
        report(f"spatial_wgrad fused T{T} {cin}->{cout}", ms, fl, 4.0 * rows * (cin + cout))

        def pair():
            block.mix_agg(x, agg, a, cin)
            ops.rows_wgrad(agg, dy, K=3 * cin, N=cout)
        ms = timeit(pair, reps)
        report(f"mix_agg + rows_wgrad     T{T} {cin}->{cout}", ms, fl, 4.0 * rows * (8 * cin + cout))


def bench_spatial_bwd(B, reps):
    """Backward of the spatial stage: the fused tile kernel (dagg on chip) vs pw_gemm(dy . Wd) / rows_gemm + joint_dagg."""
    for T, cin, cout in ((300, 64, 64), (300, 64, 128), (150, 128, 128), (150, 128, 256), (75, 256, 256)):
        x, a, dy = rnd(B, T, V, cin), rnd(B, 3, V, V) * 0.2, rnd(B, T, V, cout)
        dx = rnd(B, T, V, cin)
        rows = B * T * V
        fl = rows * (12.0 * V * cin + 6.0 * cin * cout)
        wt = rnd(1, cout, 3 * cin) * cout ** -0.5
        if ops.spatial_bwd_tile_available(V, cin, cout):
            w3 = ops.pack_split3(wt)
            ms = timeit(lambda: ops.spatial_bwd_tile(dy, x, a, w3, dx, accumulate=True), reps)
            report(f"spatial_bwd_tile (fused) T{T} {cin}->{cout}", ms, fl, 4.0 * rows * (3 * cin + cout))
        dagg = torch.empty(B, T, V, 3 * cin, device=DEV)
        if ops.pw_gemm_available():
            w3 = ops.pack_split3(wt)
            ms1 = timeit(lambda: ops.pw_gemm(dy, w3, dagg), reps)
        else:
            ms1 = timeit(lambda: ops.rows_gemm(dy, wt, dagg, K=cout, N=3 * cin), reps)
        ms1r = timeit(lambda: ops.rows_gemm(dy, wt, dagg, K=cout, N=3 * cin), reps)
        ms2 = timeit(lambda: ops.joint_dagg(x, dagg, a, dx, accumulate=True), reps)
        report(f"  pw_gemm {ms1:.3f} (f32 row GEMM {ms1r:.3f}) + joint_dagg {ms2:.3f}  T{T} {cin}->{cout}", min(ms1, ms1r) + ms2, fl,
               4.0 * rows * (9 * cin + cout))


def bench_emb_fwd(B, reps):
    """Forward of the attention embeddings: the tile kernel (gram on chip) vs the 1x1 product + joint_gram, at the headline model's
    (T, cin, ic) of blocks l1-l3, l4, l5-l6, l7, l8-l9."""
    for T, cin, ic in ((300, 64, 16), (300, 64, 32), (150, 128, 32), (150, 128, 64), (75, 256, 64)):
        ce = 6 * ic
        x, bias = rnd(B, T, V, cin), rnd(ce)
        wt = rnd(1, cin, ce) * cin ** -0.5
        rows = B * T * V
        fl = rows * (2.0 * cin * ce + 2.0 * V * 3 * ic)
        tag = f"T{T} cin {cin} ic {ic}"
        if ops.emb_fwd_tile_available(V, ic, cin):
            w3 = ops.pack_split3(wt)
            ms = timeit(lambda: ops.emb_fwd_tile(x, w3, bias, ic=ic), reps)
            report(f"emb_fwd_tile             {tag}", ms, fl, 4.0 * rows * (cin + ce))
        emb = torch.empty(B, T, V, ce, device=DEV)
        if ops.pw_gemm_available() and cin >= 128:
            w3 = ops.pack_split3(wt)
            ms1 = timeit(lambda: ops.pw_gemm(x, w3, emb, bias=bias), reps)
        else:
            ms1 = timeit(lambda: ops.rows_gemm(x, wt, emb, K=cin, N=ce, bias=bias), reps)
        ms2 = timeit(lambda: ops.joint_gram(emb, emb, [(2 * k * ic, (2 * k + 1) * ic, ic) for k in range(3)]), reps)
        report(f"  1x1 product {ms1:.3f} + joint_gram {ms2:.3f}  {tag}", ms1 + ms2, fl, 4.0 * rows * (cin + 2 * ce))


def bench_emb_bwd(B, reps):
    """Backward of the attention embeddings: the two tile kernels (demb on chip) vs joint_mix_vec(demb) + the 1x1 data gradient + the 1x1
    weight gradient, at the headline model's (T, cin, ic) of blocks l1-l3, l4, l5-l6, l7, l8-l9."""
    for T, cin, ic in ((300, 64, 16), (300, 64, 32), (150, 128, 32), (150, 128, 64), (75, 256, 64)):
        ce = 6 * ic
        emb, x, ds = rnd(B, T, V, ce), rnd(B, T, V, cin), rnd(B, 3, V, V) * 0.2
        dx, demb = rnd(B, T, V, cin), torch.empty(B, T, V, ce, device=DEV)
        wt = rnd(1, ce, cin) * ce ** -0.5
        rows = B * T * V
        fl_mix, fl_mm = rows * 2.0 * V * ce, rows * 2.0 * ce * cin
        tag = f"T{T} cin {cin} ic {ic}"
        if ops.emb_tile_available(V, ic, cin):
            w3 = ops.pack_split3(wt)
            ms_a = timeit(lambda: ops.emb_dx_tile(emb, ds, w3, dx, ic=ic, accumulate=True), reps)
            report(f"emb_dx_tile              {tag}", ms_a, fl_mix + fl_mm, 4.0 * rows * (ce + 2 * cin))
            ms_b = timeit(lambda: ops.emb_wgrad_tile(emb, x, ds, ic=ic), reps)
            report(f"emb_wgrad_tile           {tag}", ms_b, fl_mix + fl_mm, 4.0 * rows * (ce + cin))
            # operands that are NOT resident in the Infinity Cache (what the step sees): a ring of input sets > 256 MB
            nset = max(2, int(1.2e9 / (4.0 * rows * (ce + 2 * cin))))
            ring = [(rnd(B, T, V, ce), rnd(B, T, V, cin), rnd(B, T, V, cin)) for _ in range(nset)]
            it = [0]

            def cold():
                e_, x_, d_ = ring[it[0] % nset]
                it[0] += 1
                ops.emb_dx_tile(e_, ds, w3, d_, ic=ic, accumulate=True)
                ops.emb_wgrad_tile(e_, x_, ds, ic=ic)
            ms_c = timeit(cold, max(reps, 2 * nset))
            report(f"  both, operands from HBM (ring of {nset} sets; warm {ms_a + ms_b:.3f})", ms_c, 2 * (fl_mix + fl_mm), 4.0 * rows * (2 * ce + 3 * cin))
            del ring
        ms1 = timeit(lambda: block.mix_demb(emb, demb, ds, ic), reps)
        if ops.pw_gemm_available():
            w3 = ops.pack_split3(wt)
            ms2 = min(timeit(lambda: ops.pw_gemm(demb, w3, dx, accumulate=True), reps),
                      timeit(lambda: ops.rows_gemm(demb, wt, dx, K=ce, N=cin, accumulate=True), reps))
        else:
            ms2 = timeit(lambda: ops.rows_gemm(demb, wt, dx, K=ce, N=cin, accumulate=True), reps)
        ms3 = timeit(lambda: ops.rows_wgrad(x, demb, K=cin, N=ce), reps)
        report(f"  mix_demb {ms1:.3f} + 1x1 dgrad {ms2:.3f} + 1x1 wgrad {ms3:.3f}  {tag}", ms1 + ms2 + ms3, fl_mix + 2 * fl_mm,
               4.0 * rows * (4 * ce + 3 * cin))


def bench_joint(B, reps):
    for order in ((1,), (2, 1)):
        ops.paths().mix_vw_order = order
        print(f"-- channel-group mix kernels, channels per lane preference {order}")
        for T, c in ((300, 64), (150, 128), (75, 256)):
            ic = c // 4
            x, a = rnd(B, T, V, c), rnd(B, 3, V, V) * 0.2
            agg, dx = torch.empty(B, T, V, 3 * c, device=DEV), torch.zeros(B, T, V, c, device=DEV)
            rows = B * T * V
            ms = timeit(lambda: block.mix_agg(x, agg, a, c), reps)
            report(f"mix_agg T{T} C{c}", ms, 6.0 * rows * V * c, 4.0 * rows * 4 * c)
            ms = timeit(lambda: block.mix_dx(agg, dx, a, c, accumulate=True), reps)
            report(f"mix_dx (+=) T{T} C{c}", ms, 6.0 * rows * V * c, 4.0 * rows * 5 * c)
            ms = timeit(lambda: block.mix_dx(agg, dx, a, c, accumulate=False), reps)
            report(f"mix_dx (=)  T{T} C{c}", ms, 6.0 * rows * V * c, 4.0 * rows * 4 * c)
            emb, demb = rnd(B, T, V, 6 * ic), torch.empty(B, T, V, 6 * ic, device=DEV)
            ms = timeit(lambda: block.mix_demb(emb, demb, a, ic), reps)
            report(f"mix_demb T{T} ic{ic}", ms, 2.0 * rows * V * 6 * ic, 4.0 * rows * 12 * ic)
    ops.paths().mix_vw_order = (2, 1)
    for T, c in ((300, 64), (150, 128), (75, 256)):
        ic = c // 4
        rows = B * T * V
        x, agg, emb = rnd(B, T, V, c), rnd(B, T, V, 3 * c), rnd(B, T, V, 6 * ic)
        ms = timeit(lambda: ops.joint_gram(emb, emb, [(2 * k * ic, (2 * k + 1) * ic, ic) for k in range(3)]), reps)
        report(f"gram score T{T} ic{ic}", ms, 6.0 * rows * V * ic, 4.0 * rows * 6 * ic)
        ms = timeit(lambda: ops.joint_gram(x, agg, [(0, k * c, c) for k in range(3)]), reps)
        report(f"gram dA^ T{T} C{c}", ms, 6.0 * rows * V * c, 4.0 * rows * 4 * c)
        a, dx = rnd(B, 3, V, V) * 0.2, torch.zeros(B, T, V, c, device=DEV)
        ms = timeit(lambda: ops.joint_dagg(x, agg, a, dx, accumulate=True), reps)
        report(f"joint_dagg (dx += and dA^ fused) T{T} C{c}", ms, 12.0 * rows * V * c, 4.0 * rows * 6 * c)


def bench_elem(B, reps):
    for T, c in ((300, 64), (75, 256)):
        rows = B * T * V
        a, b = rnd(B, T, V, c), rnd(B, T, V, c)
        vec = torch.stack([torch.zeros(c), torch.ones(c), torch.ones(c), torch.zeros(c)]).to(DEV).contiguous()
        out = torch.empty_like(a)
        ms = timeit(lambda: ops.bn_act(a, vec, b, None, relu=True, out=out), reps)
        report(f"bn_act identity T{T} C{c}", ms, 0, 4.0 * rows * 3 * c)
        ms = timeit(lambda: ops.bn_act_bwd(a, out, b, vec, a, None, res_mode=1, db=torch.empty_like(a)), reps)
        report(f"bn_act_bwd identity T{T} C{c} (reduce+apply)", ms, 0, 4.0 * rows * 8 * c)
        ms = timeit(lambda: ops.bn_act(a, vec, b, None, relu=True, out=out, sign_mask=True), reps)
        report(f"bn_act identity + sign bits T{T} C{c}", ms, 0, 4.0 * rows * (3 + 1 / 32) * c)
        _, sign = ops.bn_act(a, vec, b, None, relu=True, out=out, sign_mask=True)
        ms = timeit(lambda: ops.bn_act_bwd(a, None, b, vec, a, None, res_mode=1, db=torch.empty_like(a), sign_mask=sign), reps)
        report(f"bn_act_bwd identity T{T} C{c} (reduce+apply, gate from sign bits)", ms, 0, 4.0 * rows * (6 + 2 / 32) * c)
        ms = timeit(lambda: ops.col_sum(a, c), reps)
        report(f"col_sum T{T} C{c}", ms, 0, 4.0 * rows * c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="gemm,pw,tconv,wgrad,spatial,spatial_wgrad,spatial_bwd,emb_fwd,emb_bwd,joint,elem")
    ap.add_argument("--tune", default="", help="fgcn_set_tuning pairs, e.g. 5=1,4=1")
    ap.add_argument("--math", default="f32", choices=("f32", "bf16", "bf16x3", "f16x2"), help="fgcn_set_math_mode")
    args = ap.parse_args()
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        _lib.load().fgcn_set_tuning(int(k), int(v))
        TUNE[int(k)] = int(v)
        print(f"-- tuning {k} = {v}")
    ops.set_math_mode(args.math)
    print(f"-- math mode {args.math}")
    fns = dict(gemm=bench_gemm, pw=bench_pw, tconv=bench_tconv, wgrad=bench_wgrad, spatial=bench_spatial, spatial_wgrad=bench_spatial_wgrad, spatial_bwd=bench_spatial_bwd, emb_fwd=bench_emb_fwd, emb_bwd=bench_emb_bwd, joint=bench_joint, elem=bench_elem)
    for k in args.only.split(","):
        fns[k](args.b, args.reps)


if __name__ == "__main__":
    main()
