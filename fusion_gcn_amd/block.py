"""One AGCN / ST-GCN block (SpatialTemporalConv) on the HIP kernels: forward, backward, autograd glue.

Reference semantics: torch_src/models/mmargcn/agcn.py:96-115 (SpatialGraphConv.forward), :49-51 (TemporalConv),
:134-136 (SpatialTemporalConv.forward) and the autograd backward of that graph (formulas: SURVEY.md Appendix A).

Internal layout is channels-last (B, T, V, C) with C padded to a multiple of 4 (only the 3-channel network input
needs padding).  Every arithmetic step is a libfgcn kernel (fusion_gcn_amd/ops.py); torch is used for buffers,
the stream and trivial weight re-layout (cat / permute of the small parameter tensors, cached per parameter version).

Kernel schedule of one block, train mode (B = N*M samples):
  forward   emb_fwd_tile [split modes: theta|phi 1x1 with the V x V affinity gram formed from the tile on chip]
               (else: rows_gemm / pw_gemm(theta|phi 1x1) -> joint_gram) -> adj_softmax (A^ = A + B + C)
            spatial_fwd  [fused x.A^_k + conv_d, BN partial sums]   (or joint_mix + rows_gemm when fused_spatial=False)
            bn_finalize -> [rows_gemm(down) -> bn_finalize] -> bn_act (BN + down/identity + ReLU = G)
            tconv_halo (9x1 temporal conv, stride 1; rows_gemm for stride 2) with BN partial sums -> bn_finalize
            [rows_gemm(residual 1x1 stride s) -> bn_finalize] -> bn_act (BN + residual + ReLU = O)
  backward  bn_act_bwd (O)  -> tconv_halo(data gradient 9x1; two parity passes when strided) / tconv_wgrad(all taps)
            -> bn_act_bwd (G)
            spatial_wgrad_tile [bf16x3, channels in 64s: conv_d's weight gradient, aggregation on chip, whole frame tiles]
               (else: spatial_wgrad up to 128 outputs; beyond: joint_mix_vec(agg) -> pw_wgrad)
            -> spatial_bwd_tile [bf16x3, >= 64 inputs: dagg = dY.Wd on chip, dx and dA^ in one launch]
               (else: pw_gemm / rows_gemm(dY.Wd) -> joint_dagg(dx, dA^))
            -> adj_softmax_bwd -> emb_dx_tile + emb_wgrad_tile [split modes, channels in 64s: the embedding gradient on chip]
               (else: joint_mix_vec(dtheta, dphi) -> pw_gemm / rows_gemm(dx) / rows_wgrad(theta|phi))
            + down / residual conv dgrad & wgrad, bias gradients by col_sum; every weight gradient is reduced from its
            slabs straight into the parameter's (out, in, taps, 1) layout (reduce_sum_strided).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import ops
from .packing import Form, PackedWeights, Seg

NUM_SUBSETS = 3


def _r4(c: int) -> int:
    return (c + 3) // 4 * 4


@dataclass(frozen=True)
class BlockConfig:
    cin: int
    cout: int
    stride: int = 1
    residual: str = "identity"      # "none" | "identity" | "conv"
    has_down: bool = False
    static_adjacency: bool = False  # ST-GCN special case: A^ = A + B, no data-dependent C_k
    fused_spatial: bool = True      # north-star fused kernel vs. joint_mix + rows_gemm

    @property
    def ic(self) -> int:
        return self.cout // 4

    @property
    def cx(self) -> int:            # padded input channels
        return _r4(self.cin)

    def validate(self) -> None:
        if self.cout % 64 != 0:
            raise ValueError(f"HIP AGCN block needs out_channels % 64 == 0 (got {self.cout}): the embedding "
                             "channels (out_channels/4 per subset) are tiled in groups of 16")
        if self.residual == "identity" and not (self.cin == self.cout and self.stride == 1):
            raise ValueError("identity residual needs cin == cout and stride 1")
        if not self.has_down and self.cin != self.cout:
            raise ValueError("cin != cout needs the down branch")


# parameter order of a block (autograd.Function inputs); bn buffers travel separately
def param_names(cfg: BlockConfig) -> List[str]:
    names = ["gcn1.adj_b"]
    for grp in ("conv_a", "conv_b", "conv_d"):
        for k in range(NUM_SUBSETS):
            names += [f"gcn1.{grp}.{k}.weight", f"gcn1.{grp}.{k}.bias"]
    names += ["gcn1.bn.weight", "gcn1.bn.bias"]
    if cfg.has_down:
        names += ["gcn1.down.0.weight", "gcn1.down.0.bias", "gcn1.down.1.weight", "gcn1.down.1.bias"]
    names += ["tcn1.conv.weight", "tcn1.conv.bias", "tcn1.bn.weight", "tcn1.bn.bias"]
    if cfg.residual == "conv":
        names += ["residual.conv.weight", "residual.conv.bias", "residual.bn.weight", "residual.bn.bias"]
    return names


def bn_names(cfg: BlockConfig) -> List[str]:
    out = ["gcn1.bn"]
    if cfg.has_down:
        out.append("gcn1.down.1")
    out.append("tcn1.bn")
    if cfg.residual == "conv":
        out.append("residual.bn")
    return out


# ---- weight re-layout (tiny tensors; torch plumbing) ---------------------------------------------------------------
def _pad_last(t: torch.Tensor, n: int) -> torch.Tensor:
    return t if t.shape[-1] == n else F.pad(t, (0, n - t.shape[-1]))


def pack_weights(P: Dict[str, torch.Tensor], cfg: BlockConfig) -> PackedWeights:
    """Reference-layout parameters -> the packed forms rows_gemm / spatial_fwd / tconv_halo stream, as DATA (packing.Form: the
    logical (taps, K, N) matrix as windows of the parameter tensors + the consumer's layout); a form is materialised on first use
    and refreshed with all the others in one launch afterwards.  The input-channel dimension is zero-padded from cin to cx (= cin
    rounded up to 4): the kernels then see a cx-channel block whose extra input channel is identically zero (only the 3-channel
    network input pads).  The math mode is read once, here (the block keys its set of forms on it)."""
    cin, cout, ic, cx = cfg.cin, cfg.cout, cfg.ic, cfg.cx
    mode = ops.get_math_mode()
    x3 = mode in ops.SPLIT_MODES                     # modes whose halo kernel takes the split weight form
    split_form = "split2h" if mode == "f16x2" else "split3"      # f16x2: block-scaled two-way f16 split (FGCN_PACK_SPLIT2H)
    conv_form = split_form if x3 else "k4"
    F: Dict[str, Form] = {}
    conv = lambda name: P[name].detach()             # noqa: E731  (O, I, kt, 1) contiguous: strides o: I*kt, i: kt, tap: 1

    def one_by_one(name, K_in):
        """(1, K_in-padded, O) and its transpose from a 1x1 conv weight (O, K_in, 1, 1)"""
        w = conv(name)
        o = w.shape[0]
        return ([Seg(w, st_k=1, st_n=K_in, klen=K_in, nlen=o)], [Seg(w, st_k=K_in, st_n=1, klen=o, nlen=K_in)])

    if not cfg.static_adjacency:
        # embedding channel order [th0 ph0 th1 ph1 th2 ph2]
        emb = [conv(f"gcn1.conv_{g}.{k}.weight") for k in range(NUM_SUBSETS) for g in "ab"]
        F["emb"] = Form("plain", 1, cx, 6 * ic, [Seg(w, st_k=1, st_n=cin, klen=cin, nlen=ic, n0=j * ic) for j, w in enumerate(emb)])
        F["emb_t"] = Form("plain", 1, 6 * ic, cx, [Seg(w, st_k=cin, st_n=1, klen=ic, nlen=cin, k0=j * ic) for j, w in enumerate(emb)])
        F["emb_b"] = Form("plain", 1, 1, 6 * ic, [Seg(P[f"gcn1.conv_{g}.{k}.bias"].detach(), st_k=0, st_n=1, klen=1, nlen=ic, n0=(2 * k + j) * ic)
                                                   for k in range(NUM_SUBSETS) for j, g in enumerate("ab")], shape=(6 * ic,))
    wd = [conv(f"gcn1.conv_d.{k}.weight") for k in range(NUM_SUBSETS)]
    d_rows = [Seg(w, st_k=1, st_n=cin, klen=cin, nlen=cout, k0=k * cx) for k, w in enumerate(wd)]          # (3cx, cout)
    F["d"] = Form("plain", 1, 3 * cx, cout, d_rows, shape=(3 * cx, cout))
    if mode in ops.X3_MODES and cx % 32 == 0:        # the fused spatial kernel's form (ops.pack_spatial)
        F["d4"] = Form("split2h_acc" if mode == "f16x2" else "split3_acc", 1, 3 * cx, cout, d_rows)
    else:
        F["d4"] = Form("k4", 1, 3 * cx, cout, d_rows, shape=(3 * cx // 4, cout, 4))
    if mode in ("bf16x3", "bf16") and cx % 64 == 0:  # the tile form of the fused spatial kernel (ops.spatial_fwd_tile): plain split form (bf16: its part 0)
        F["d_s3"] = Form("split3", 1, 3 * cx, cout, d_rows)
    d_cols = [Seg(w, st_k=cin, st_n=1, klen=cout, nlen=cin, n0=k * cx) for k, w in enumerate(wd)]          # (1, cout, 3cx)
    F["d_t"] = Form("plain", 1, cout, 3 * cx, d_cols)
    # the kernel adds the sum of the three biases: overlapping segments are summed
    F["d_b"] = Form("plain", 1, 1, cout, [Seg(P[f"gcn1.conv_d.{k}.bias"].detach(), st_k=0, st_n=1, klen=1, nlen=cout)
                                          for k in range(NUM_SUBSETS)], shape=(cout,))
    if cfg.has_down:
        fwd, bwd = one_by_one("gcn1.down.0.weight", cin)
        F["down"], F["down_t"] = Form("plain", 1, cx, cout, fwd), Form("plain", 1, cout, cx, bwd)
    wt = conv("tcn1.conv.weight")                    # (o, c, kt, 1)
    kt = wt.shape[2]
    t_seg = lambda **kw: [Seg(wt, st_tap=1, st_k=kt, st_n=cout * kt, klen=cout, nlen=cout, **kw)]           # noqa: E731  (kt, c, o)
    tt_seg = lambda **kw: [Seg(wt, st_tap=1, st_k=cout * kt, st_n=kt, klen=cout, nlen=cout, **kw)]          # noqa: E731  (kt, o, c)
    F["t"] = Form("plain", kt, cout, cout, t_seg(tlen=kt))
    F["t_t"] = Form("plain", kt, cout, cout, tt_seg(tlen=kt))
    # the halo-tile kernel's forms (ops.pack_conv: k-interleaved f32, or the three-way bf16 split in the bf16 math modes); a
    # stride-2 conv runs as an even-tap and an odd-tap pass
    if cfg.stride == 1:
        F["t4"] = Form(conv_form, kt, cout, cout, t_seg(tlen=kt))
        F["t_t4"] = Form(conv_form, kt, cout, cout, tt_seg(tlen=kt))
    else:
        for par, tag in ((0, "e"), (1, "o")):        # data gradient; forward too in the split modes (see temporal_fwd)
            n_par = (kt - par + 1) // 2
            F[f"t_t4_{tag}"] = Form(conv_form, n_par, cout, cout, tt_seg(tlen=n_par, tap0=par, tap_step=2))
            if x3:
                F[f"t4_{tag}"] = Form(conv_form, n_par, cout, cout, t_seg(tlen=n_par, tap0=par, tap_step=2))
    if cfg.residual == "conv":
        fwd, bwd = one_by_one("residual.conv.weight", cin)
        F["res"], F["res_t"] = Form("plain", 1, cx, cout, fwd), Form("plain", 1, cout, cx, bwd)
    if x3:                                           # split forms of the 1x1 weights pw_gemm may route to the halo kernel
        for key in ("emb", "emb_t", "d_t", "down", "down_t"):
            if key in F and F[key].K % 32 == 0:
                F[key + "_s3"] = Form(split_form, 1, F[key].K, F[key].N, F[key].segs)
    # the fused spatial backward (ops.spatial_bwd_tile) multiplies three-way bf16 splits in every float32-class mode: with the f16x2
    # products the block keeps that form of d_t beside the two-way f16 one
    if mode in ("bf16x3", "f16x2", "bf16") and cx % 64 == 0 and cout % 64 == 0:
        F["d_t_b3"] = Form("split3", 1, cout, 3 * cx, d_cols)      # (a form of its own: forms are materialised per key; bf16 reads its part 0)
    # the embedding backward in tile form (ops.emb_dx_tile: the embedding gradient on chip) streams the three-way bf16 split of emb_t in
    # every split mode (bf16: its part 0)
    if "emb_t" in F and mode in ops.SPLIT_MODES and cx % 64 == 0:
        F["emb_t_b3"] = Form("split3", 1, 6 * ic, cx, F["emb_t"].segs)
    # ... and the embedding forward with the gram on chip (ops.emb_fwd_tile) the split of emb
    if "emb" in F and mode in ops.SPLIT_MODES and cx % 32 == 0:
        F["emb_b3"] = Form("split3", 1, cx, 6 * ic, F["emb"].segs)
    return PackedWeights(F, P["tcn1.conv.weight"].device)


# Which kernel form a stage takes (tile forms, fusions, thresholds) is a per-context option: fusion_gcn_amd/paths.py carries the fields,
# their defaults and the measurement behind each; `ops.paths()` is the calling thread's current set (the backward of a block runs in
# its forward's context, so both halves see the same options).
def _pw_min_k(rows: int) -> int:
    """contraction depth from which a 1x1 convolution with a split form goes to the persistent split-bf16 row GEMM (ops.pw_gemm,
    fgcn_pw.hip); below it (and in math mode f32) the exact-f32 row GEMM runs (paths.PathOptions.pw_min_k)"""
    o = ops.paths()
    if ops.get_math_mode() == "f16x2":
        return o.pw_min_k_f16x2
    return min(o.pw_min_k, 64) if rows < o.pw_small_rows else o.pw_min_k


def pw_routed(W, key: str, x: torch.Tensor, K: int) -> bool:
    """Whether pw_gemm sends this 1x1 convolution to the persistent split row GEMM (the kernel that can record max |x|)."""
    return (key + "_s3") in W and K % 32 == 0 and x.shape[3] == K and K >= _pw_min_k(x.numel() // x.shape[3])


def pw_gemm(x: torch.Tensor, W: Dict[str, torch.Tensor], key: str, out: torch.Tensor, *, K: int, N: int,
            bias: Optional[torch.Tensor] = None, stats: bool = False, accumulate: bool = False, amax_out: Optional[torch.Tensor] = None):
    """1x1 convolution over all rows: in the split-bf16 math modes (the packed set then holds the split form of the weight) the
    persistent split-bf16 row GEMM; the exact-f32 row GEMM otherwise.  ``amax_out``: see ops.pw_gemm (ignored by the f32 kernel)."""
    w3 = W.get(key + "_s3")
    if w3 is not None and K % 32 == 0 and x.shape[3] == K and K >= _pw_min_k(x.numel() // x.shape[3]):
        return ops.pw_gemm(x, w3, out, bias=bias, stats=stats, accumulate=accumulate, amax_out=amax_out)
    return ops.rows_gemm(x, W[key], out, K=K, N=N, bias=bias, stats=stats, accumulate=accumulate)


# ---- joint-mix item tables -----------------------------------------------------------------------------------------
def spec_agg(cin: int) -> List[dict]:
    """agg[(k, c)] = sum_v x[v, c] A^_k[v, w]: out joint = column index of A^_k -> transpose = 1."""
    out = []
    for k in range(NUM_SUBSETS):
        for c0 in range(0, cin, 32):
            out.append(dict(out_c=k * cin + c0, width=min(32, cin - c0), terms=[(k, 1, c0, c0 + 16, 3)]))
    return out


def spec_dx(cin: int) -> List[dict]:
    """dx[v, c] = sum_k sum_w dagg[(k, c), w] A^_k[v, w]."""
    return [dict(out_c=c0, width=min(32, cin - c0),
                 terms=[(k, 0, k * cin + c0, k * cin + c0 + 16, 3) for k in range(NUM_SUBSETS)])
            for c0 in range(0, cin, 32)]


def _vec_width(c: int, order: Sequence[int] = None) -> int:
    """Channels per lane (1/2) for a group-of-``c``-channels mix: groups of 32*vw channels must tile ``c`` exactly
    (first such vw in preference order), or ``c`` is a single narrower group (smallest such vw: most lanes busy).
    0: no such tiling, use the dword kernel with 16-lane masks."""
    order = order or ops.paths().mix_vw_order
    for vw in order:
        if c % (32 * vw) == 0:
            return vw
    narrow = [vw for vw in order if c % vw == 0 and c < 32 * vw]
    return min(narrow) if narrow else 0


def mix_agg(x: torch.Tensor, agg: torch.Tensor, a_hat: torch.Tensor, cin: int, amax_out=None) -> bool:
    """agg[(k, c)] = x . A^_k for the three subsets (items of one channel group are adjacent: x is loaded once).  ``amax_out``:
    records max |agg| (ops.joint_mix_vec); -> whether it was recorded."""
    vw = _vec_width(cin)
    if not vw:
        ops.joint_mix(x, agg, a_hat, spec_agg(cin), in_channels=cin, out_channels=3 * cin)
        return False
    g = 32 * vw
    spec = [dict(out_c=k * cin + c0, nch=min(g, cin), terms=[(k, 1, c0)])
            for c0 in range(0, cin, g) for k in range(NUM_SUBSETS)]
    ops.joint_mix_vec(x, agg, a_hat, spec, vw=vw, amax_out=amax_out)
    return amax_out is not None


def mix_dx(dagg: torch.Tensor, dx: torch.Tensor, a_hat: torch.Tensor, cin: int, accumulate: bool) -> None:
    """dx (+)= sum_k dagg_k . A^_k^T."""
    vw = _vec_width(cin)
    if not vw:
        ops.joint_mix(dagg, dx, a_hat, spec_dx(cin), in_channels=3 * cin, out_channels=cin, accumulate=accumulate)
        return
    g = 32 * vw
    spec = [dict(out_c=c0, nch=min(g, cin), terms=[(k, 0, k * cin + c0) for k in range(NUM_SUBSETS)])
            for c0 in range(0, cin, g)]
    ops.joint_mix_vec(dagg, dx, a_hat, spec, vw=vw, accumulate=accumulate)


def mix_demb(emb: torch.Tensor, demb: torch.Tensor, d_s: torch.Tensor, ic: int) -> torch.Tensor:
    """dtheta_k = dS_k . phi_k, dphi_k = dS_k^T . theta_k over the embedding layout [th0 ph0 th1 ph1 th2 ph2];
    returns the column sums of demb (the theta|phi bias gradient), fused into the mix where the kernel allows."""
    vw = _vec_width(ic)
    if not vw:
        ops.joint_mix(emb, demb, d_s, spec_demb(ic), in_channels=6 * ic, out_channels=6 * ic)
        return ops.col_sum(demb, 6 * ic)
    g = 32 * vw
    spec = []
    for k in range(NUM_SUBSETS):
        th, ph = 2 * k * ic, (2 * k + 1) * ic
        for c0 in range(0, ic, g):
            spec.append(dict(out_c=th + c0, nch=min(g, ic), terms=[(k, 0, ph + c0)]))
            spec.append(dict(out_c=ph + c0, nch=min(g, ic), terms=[(k, 1, th + c0)]))
    if len(spec) > ops.MIX_MAX_ITEMS:
        ops.joint_mix_vec(emb, demb, d_s, spec, vw=vw)
        return ops.col_sum(demb, 6 * ic)
    _, sums = ops.joint_mix_vec(emb, demb, d_s, spec, vw=vw, colsum=True)
    return sums


def temporal_fwd_records_amax(W, kt: int, s: int, T: int) -> bool:
    """Whether temporal_fwd takes a halo-kernel route for these sizes (the only routes that record max |g| in math mode f16x2)."""
    pad = (kt - 1) // 2
    return (s == 1 and "t4" in W) or (s == 2 and "t4_e" in W and pad % 2 == 0 and T > 1)


def temporal_dgrad_records_amax(W, kt: int, s: int) -> bool:
    """The same for temporal_dgrad and max |du|."""
    pad = (kt - 1) // 2
    return (s == 1 and "t_t4" in W) or (s == 2 and "t_t4_e" in W and pad % 2 == 0)


def temporal_fwd(g: torch.Tensor, u: torch.Tensor, W: Dict[str, torch.Tensor], bias: torch.Tensor, kt: int, s: int,
                 stats: bool, fuse_in=None, amax_out=None):
    """u = Conv(kt x 1, stride s, pad (kt-1)//2)(g) + bias, with BatchNorm partial sums of u when ``stats``.
    ``fuse_in = (vec, shortcut, g_out, g_sign)`` (stride 1, split-bf16 kernel): the first argument is the BatchNorm input y and
    g = relu(BatchNorm(y) + shortcut) is formed inside the conv (ops.tconv_halo)."""
    pad = (kt - 1) // 2
    T, Tp = g.shape[1], u.shape[1]
    if s == 1 and "t4" in W:
        return ops.tconv_halo(g, W["t4"], u, Th=T, taps=kt, tb=1, tc=-pad, bias=bias, stats=stats, fuse_in=fuse_in, amax_out=amax_out)
    assert fuse_in is None
    if s == 2 and "t4_e" in W and pad % 2 == 0 and T > 1:
        # output frame to meets tap j = 2j' + par at input frame 2 (to + j' - pad/2) + par: one pass over the even input
        # frames (taps 0, 2, ..), one accumulating pass over the odd ones (which also takes the BatchNorm sums)
        ops.tconv_halo(g, W["t4_e"], u, Th=Tp, taps=(kt + 1) // 2, tb=1, tc=-(pad // 2), in_view=(2, 0, (T + 1) // 2),
                       bias=bias, amax_out=amax_out)
        return ops.tconv_halo(g, W["t4_o"], u, Th=Tp, taps=kt // 2, tb=1, tc=-(pad // 2), in_view=(2, 1, T // 2),
                              stats=stats, accumulate=True, amax_out=amax_out)
    # strided forward in f32: the per-tap row GEMM measures faster than two accumulating halo passes over the even / odd
    # input frames (1.26 vs 1.54 ms at 128 channels, T 300 -> 150); on the split-bf16 kernels (math mode bf16x3: the packed
    # set then holds t4_e / t4_o) the two passes win
    return ops.rows_gemm(g, W["t"], u, K=g.shape[3], N=u.shape[3], tmap=ops.conv_tmap(kt, s), bias=bias, stats=stats)


def temporal_dgrad(du: torch.Tensor, dg: torch.Tensor, W: Dict[str, torch.Tensor], kt: int, s: int, bn_bwd=None, amax_out=None):
    """dg = data gradient of that convolution: dg[t] = sum_j W_j^T du[(t + pad - j) / s].  ``bn_bwd`` (stride 1, split-bf16 kernel):
    the BatchNorm-backward sums of dg against (a, sign image, vec) from the kernel's epilogue -> partials, else None."""
    pad = (kt - 1) // 2
    T, Tp = dg.shape[1], du.shape[1]
    if s == 1 and "t_t4" in W:
        return ops.tconv_halo(du, W["t_t4"], dg, Th=T, taps=kt, tb=-1, tc=pad, bn_bwd=bn_bwd, amax_out=amax_out)
    elif s == 2 and "t_t4_e" in W and pad % 2 == 0:
        # frame t = 2*th + par only meets taps j = 2j' + par, at du frame th + pad/2 - j'
        ops.tconv_halo(du, W["t_t4_e"], dg, Th=(T + 1) // 2, taps=(kt + 1) // 2, tb=-1, tc=pad // 2, out_view=(2, 0), amax_out=amax_out)
        if T > 1:
            ops.tconv_halo(du, W["t_t4_o"], dg, Th=T // 2, taps=kt // 2, tb=-1, tc=pad // 2, out_view=(2, 1), amax_out=amax_out)
    else:
        ops.rows_gemm(du, W["t_t"], dg, K=du.shape[3], N=dg.shape[3], tmap=ops.conv_dgrad_tmap(kt, s))


def spec_demb(ic: int) -> List[dict]:
    """dtheta_k = dS_k . phi_k (transpose 0), dphi_k = dS_k^T . theta_k (transpose 1); embedding channels are
    [th0 ph0 th1 ph1 th2 ph2], each ``ic`` (a multiple of 16) wide, so a 32-lane tile may straddle two groups."""
    def source(c):                       # 16-channel group starting at c -> (mat, transpose, source channel)
        grp, off = divmod(c, ic)
        k, is_phi = divmod(grp, 2)
        return (k, 1, 2 * k * ic + off) if is_phi else (k, 0, (2 * k + 1) * ic + off)
    out = []
    for c0 in range(0, 6 * ic, 32):
        lo, hi = source(c0), source(c0 + 16)
        if lo[:2] == hi[:2]:
            terms = [(lo[0], lo[1], lo[2], hi[2], 3)]
        else:
            terms = [(lo[0], lo[1], lo[2], lo[2], 1), (hi[0], hi[1], hi[2], hi[2], 2)]
        out.append(dict(out_c=c0, width=32, terms=terms))
    return out


def emb_bwd_tile_ok(W, cfg: BlockConfig, B: int, T: int, V: int, cx: int, o_) -> bool:
    """Whether block_backward takes the tile form of the embedding backward (ops.emb_dx_tile + ops.emb_wgrad_tile) for this block -- one
    predicate for the backward and for the forward, which stores emb as bfloat16 in math mode bf16 only if its readers are those two."""
    cin, ic = cx, cfg.ic
    small = B * T * V * max(6 * ic, cx) * 4 < 0x7FFF0000
    return bool(o_.emb_tile and cin <= o_.get("emb_tile_max_cin", ops.get_math_mode()) and "emb_t_b3" in W and cx == cfg.cin
                and ops.emb_tile_available(V, ic, cin) and small)


def half_storage_ok(W, kt: int, s: int, T: int, train: bool, o_) -> bool:
    """Whether this block keeps G and dU in bfloat16 (paths.half_storage): math mode bf16, a training step (the eval-mode bias
    gradient reads dU as f32), and the three consumers on their bfloat16-input kernels -- the halo conv forward and data gradient
    (the routes temporal_fwd / temporal_dgrad take for these sizes) and the all-taps weight gradient (tap counts it is built for)."""
    if not (train and ops.get_math_mode() == "bf16" and o_.get("half_storage", "bf16")) or kt <= 1:
        return False
    pad = (kt - 1) // 2
    per_pass = [kt] if s == 1 else [len([j for j in range(kt) if (j - pad) % s == par]) for par in range(s)]
    return (temporal_fwd_records_amax(W, kt, s, T) and temporal_dgrad_records_amax(W, kt, s)
            and all(n in ops.TWGRAD_TAPS_SPLIT for n in per_pass if n))


# ---- forward ---------------------------------------------------------------------------------------------------------
def _bn_vec(part, count, P, bufs, name, train):
    g, b = P[f"{name}.weight"], P[f"{name}.bias"]
    rm, rv = bufs[f"{name}.running_mean"], bufs[f"{name}.running_var"]
    if train:
        return ops.bn_finalize(part, count, g, b, rm, rv)
    return ops.bn_eval_coeffs(g, b, rm, rv)


def half_activations_on(train: bool, o_) -> bool:
    """Whether this block keeps its activation-sized tensors in bfloat16 (paths.half_activations): math mode bf16, a training step."""
    return bool(train and ops.get_math_mode() == "bf16" and o_.get("half_activations", "bf16") and o_.get("half_storage", "bf16"))


def block_forward(x: torch.Tensor, P: Dict[str, torch.Tensor], bufs: Dict[str, torch.Tensor], W: Dict[str, torch.Tensor],
                  cfg: BlockConfig, train: bool, pool_groups: int = 0, inference: bool = False, out_half: bool = False):
    """x (B, T, V, cx) -> O (B, T', V, cout); returns (O, saved-for-backward dict).  ``pool_groups`` > 0 (the model's last block): O is
    not formed, the first result is its mean over the rows of every group of B / pool_groups consecutive samples, (pool_groups, cout).
    ``inference`` (eval mode, no autograd graph): BatchNorm + shortcut + ReLU run as the EPILOGUES of the two north-star kernels where
    their inference forms exist (paths.fused_inference) -- no pre-BatchNorm tensors, no bn_act passes, nothing saved for a backward.
    Math mode bf16 (paths.half_activations): ``x`` may be a bfloat16 tensor (the previous block's output), Y / U are stored as bfloat16 where
    their producers have the form, and ``out_half`` (the caller takes a bfloat16 output: the next block) makes O one too."""
    B, T, V, cx = x.shape
    cout, ic, s = cfg.cout, cfg.ic, cfg.stride
    assert cx == cfg.cx, (cx, cfg.cx)
    cin = cx                     # kernels work on the padded channel count; the pad channel is identically zero
    Tp = (T - 1) // s + 1
    dev = x.device
    o_ = ops.paths()             # this context's kernel-form options (fusion_gcn_amd/paths.py)
    new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
    S: Dict[str, Optional[torch.Tensor]] = {"x": x}
    ha = half_activations_on(train, o_)
    x16 = x.dtype == torch.bfloat16
    _x32: List[torch.Tensor] = []

    def x32() -> torch.Tensor:      # x for a kernel without a bfloat16-input form (converted once, on first use)
        if not x16:
            return x
        if not _x32:
            _x32.append(x.float())
        return _x32[0]
    if x16 and not ha:
        raise ops._lib.FgcnError("block_forward: a bfloat16 input needs math mode bf16 with paths.half_activations (a training step)")
    # math mode f16x2: the largest magnitudes of x and G, recorded by the kernels that stage them (pw_gemm / tconv_halo), scale the
    # same tensors in the backward's weight gradients; slot 0 = x (only when the embedding runs on the split row GEMM), 1 = G
    f16x2 = ops.get_math_mode() == "f16x2"
    amax = torch.zeros(4, device=dev, dtype=torch.int32) if f16x2 else None
    S["amax"], S["x_amax"], S["g_amax"] = amax, False, False   # *_amax: the slot was really recorded (a row-GEMM fallback records nothing)

    # -- data-dependent adjacency ------------------------------------------------------------------------------------
    adj_a, adj_b = bufs["gcn1.adj_a"], P["gcn1.adj_b"].detach()
    if cfg.static_adjacency:
        emb, c_mat = None, None
        _, a_hat = ops.adj_softmax_fwd(None, 1.0, adj_a, 1, use_softmax=False, adj_b=adj_b)
    else:
        no_emb = bool(inference and not train and o_.fused_inference)      # inference: the tile form writes no embeddings at all -- it stays at every ic
        if (o_.emb_fwd_tile and cin <= o_.get("emb_fwd_tile_max_cin", ops.get_math_mode()) and (no_emb or ic <= o_.get("emb_fwd_tile_max_ic", ops.get_math_mode()))
                and "emb_b3" in W and ops.emb_fwd_tile_available(V, ic, cin)
                and B * T * V * max(cin, 6 * ic) * 4 < 0x7FFF0000):
            # emb written once, the gram from the tile on chip (inference: not written at all -- only the backward reads it)
            half_emb = bool(train and ops.get_math_mode() == "bf16" and o_.get("half_storage", "bf16") and emb_bwd_tile_ok(W, cfg, B, T, V, cx, o_))
            emb, part = ops.emb_fwd_tile(x, W["emb_b3"], W["emb_b"], ic=ic, write_emb=not no_emb, emb_bf16=half_emb)
        else:
            emb = new(B, T, V, 6 * ic)
            S["x_amax"] = f16x2 and pw_routed(W, "emb", x, cin)
            pw_gemm(x, W, "emb", emb, K=cin, N=6 * ic, bias=W["emb_b"], amax_out=amax[0:1] if S["x_amax"] else None)
            part = ops.joint_gram(emb, emb, [(2 * k * ic, (2 * k + 1) * ic, ic) for k in range(NUM_SUBSETS)])
        c_mat, a_hat = ops.adj_softmax_fwd(part, 1.0 / (ic * T), adj_a, B, adj_b=adj_b)
    S.update(emb=emb, c_mat=c_mat, a_hat=a_hat)

    # -- spatial aggregation + conv_d ------------------------------------------------------------------------------------
    infer = bool(inference and not train and o_.fused_inference and ops.inference_kernels_available())
    kt = P["tcn1.conv.weight"].shape[2]
    if infer and cfg.fused_spatial and "d_s3" in W and ops.spatial_fwd_tile_available(V, cin, cout) and (cfg.has_down or x.shape[3] >= cout):
        # north-star kernel 1, inference form: aggregation + feature contraction + BatchNorm + shortcut + ReLU in one kernel
        vec_y = _bn_vec(None, B * T * V, P, bufs, "gcn1.bn", False)
        d, vec_d = None, None
        if cfg.has_down:
            d = new(B, T, V, cout)
            pw_gemm(x32(), W, "down", d, K=cin, N=cout, bias=P["gcn1.down.0.bias"])
            vec_d = _bn_vec(None, B * T * V, P, bufs, "gcn1.down.1", False)
        g = ops.spatial_fwd_tile_bn_relu(x, a_hat, W["d_s3"], W["d_b"], vec_y, Cin=cin, Cout=cout, res=d if cfg.has_down else x, res_vec=vec_d)
        S.update(y=None, vec_y=vec_y, d=None, vec_d=vec_d, g=g, g_sign=None, half=False)
        return _temporal_stage(x, g, S, P, bufs, W, cfg, train, pool_groups, infer, kt, o_)
    # (Y as bfloat16: not when the temporal data gradient is to carry the BatchNorm-backward sums -- that epilogue reads Y as float32)
    y16 = ha and o_.half_spatial_out and not o_.get("bn_sums_in_dgrad", ops.get_math_mode())
    if cfg.fused_spatial and o_.spatial_tile and cout >= o_.get("spatial_tile_min_cout", ops.get_math_mode()) and "d_s3" in W and ops.spatial_fwd_tile_available(V, cin, cout):
        y, part = ops.spatial_fwd_tile(x if y16 else x32(), a_hat, W["d_s3"], W["d_b"], Cin=cin, Cout=cout, stats=train, y_bf16=y16)
    elif cfg.fused_spatial:
        y, part = ops.spatial_fwd(x32(), a_hat, W["d4"], W["d_b"], Cin=cin, Cout=cout, stats=train)
    else:
        agg = new(B, T, V, 3 * cin)
        mix_agg(x32(), agg, a_hat, cin)
        y = new(B, T, V, cout)
        part = ops.rows_gemm(agg, W["d"].unsqueeze(0), y, K=3 * cin, N=cout, bias=W["d_b"], stats=train)
    vec_y = _bn_vec(part, B * T * V, P, bufs, "gcn1.bn", train)
    # math mode bf16, training: G (the temporal conv's input) is stored as bfloat16 -- only bf16 MFMA staging reads it (the conv and its
    # weight gradient), so the values those kernels multiply are the same and they copy half the bytes (paths.half_storage)
    kt = P["tcn1.conv.weight"].shape[2]
    half = half_storage_ok(W, kt, s, T, train, o_)
    if cfg.has_down:
        # (paths.half_activations: the shortcut conv reads the bfloat16 x and writes a bfloat16 d -- the typed row GEMMs)
        d = torch.empty((B, T, V, cout), device=dev, dtype=torch.bfloat16) if (ha and o_.half_shortcuts) else new(B, T, V, cout)
        part = pw_gemm(x if (ha and o_.half_shortcuts) else x32(), W, "down", d, K=cin, N=cout, bias=P["gcn1.down.0.bias"], stats=train)
        vec_d = _bn_vec(part, B * T * V, P, bufs, "gcn1.down.1", train)
        g, g_sign = ops.bn_act(y, vec_y, d, vec_d, relu=True, sign_mask=True, out_bf16=half)
    else:
        d, vec_d = None, None
    # identity blocks on the split-bf16 kernels: G = relu(BatchNorm(y) + x) is formed INSIDE the temporal conv while it stages its
    # image (north-star kernel 2: "temporal 9x1 conv + BN + ReLU"), G and its sign image come out as by-products -- no bn_act pass
    fuse_g = (o_.fuse_g and not half and not ha and not cfg.has_down and s == 1 and kt > 1 and "t4" in W and ops.tconv_halo_bn_sums()
              and cx == cout and V <= 32 and (B * T * V * cout) % 8 == 0)
    if fuse_g:
        g = new(B, T, V, cout)
        g_sign = torch.empty((B * T * V * cout) // 8, device=dev, dtype=torch.uint8)
    elif not cfg.has_down:
        g, g_sign = ops.bn_act(y, vec_y, x, None, relu=True, sign_mask=True, out_bf16=half)
    if half and g_sign is None:      # (no sign image: the backward would gate on g itself, which it reads as f32 -- cout % 64 == 0 rules it out)
        raise ops._lib.FgcnError("half-precision storage of G needs the sign image (element count a multiple of 8)")
    S.update(y=y, vec_y=vec_y, d=d, vec_d=vec_d, g=g, g_sign=g_sign, half=half)   # *_sign: 1 bit per element, the backward's ReLU gate

    return _temporal_stage(x, y if fuse_g else g, S, P, bufs, W, cfg, train, pool_groups, infer, kt, o_, fuse_in=(vec_y, x, g, g_sign) if fuse_g else None,
                           x32=x32, out_half=bool(ha and out_half))


def _temporal_stage(x, g, S, P, bufs, W, cfg: BlockConfig, train: bool, pool_groups: int, infer: bool, kt: int, o_, fuse_in=None, x32=None,
                    out_half: bool = False):
    """The second half of block_forward: the temporal conv, its BatchNorm, the block's shortcut and ReLU (agcn.py:49-51,125-136)."""
    B, T, V, cx = x.shape
    cout, s = cfg.cout, cfg.stride
    cin = cx
    Tp = (T - 1) // s + 1
    dev = x.device
    new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
    f16x2 = ops.get_math_mode() == "f16x2"
    amax = S["amax"]
    if x32 is None:
        x32 = lambda: x          # noqa: E731
    if infer and s == 1 and kt > 1 and "t4" in W and not pool_groups and cfg.residual in ("none", "identity", "conv") and (cfg.residual != "identity" or x.shape[3] == cout):
        # north-star kernel 2 as the north star states it, inference form: temporal conv + BatchNorm + shortcut + ReLU in one kernel
        vec_u = _bn_vec(None, B * Tp * V, P, bufs, "tcn1.bn", False)
        r, vec_r = None, None
        if cfg.residual == "conv":
            r = new(B, Tp, V, cout)
            ops.rows_gemm(x, W["res"], r, K=cin, N=cout, tmap=(1, s, 0, 0, 1), bias=P["residual.conv.bias"])
            vec_r = _bn_vec(None, B * Tp * V, P, bufs, "residual.bn", False)
        o = new(B, Tp, V, cout)
        pad = (kt - 1) // 2
        ops.tconv_halo_bn_relu(g, W["t4"], o, taps=kt, tb=1, tc=-pad, vec=vec_u, bias=P["tcn1.conv.bias"],
                               res=x if cfg.residual == "identity" else r, res_vec=vec_r)
        S.update(u=None, vec_u=vec_u, r=None, vec_r=vec_r, o=o, o_sign=None)
        return o, S
    # paths.half_activations: U as bfloat16 where the conv that writes it has the form (the stride-1 halo kernel on a bfloat16 G; the strided
    # conv's second pass accumulates into its output and keeps float32)
    u16 = bool(half_activations_on(train, o_) and o_.half_conv_out and S.get("half") and s == 1 and kt > 1 and "t4" in W and fuse_in is None)
    u = torch.empty((B, Tp, V, cout), device=dev, dtype=torch.bfloat16) if u16 else new(B, Tp, V, cout)
    S["g_amax"] = f16x2 and temporal_fwd_records_amax(W, kt, s, T)
    part = temporal_fwd(g, u, W, P["tcn1.conv.bias"], kt, s, stats=train,
                        fuse_in=fuse_in, amax_out=amax[1:2] if S["g_amax"] else None)
    vec_u = _bn_vec(part, B * Tp * V, P, bufs, "tcn1.bn", train)
    r, vec_r = None, None
    epilogue = (lambda a, va, b, vb: ops.bn_act_pool(a, va, b, vb, pool_groups)) if pool_groups else \
               (lambda a, va, b, vb: ops.bn_act(a, va, b, vb, relu=True, sign_mask=True, out_bf16=out_half))
    if cfg.residual == "none":
        o, o_sign = epilogue(u, vec_u, None, None)
    elif cfg.residual == "identity":
        o, o_sign = epilogue(u, vec_u, x, None)
    else:
        hs = bool(half_activations_on(train, o_) and o_.half_shortcuts)
        r = torch.empty((B, Tp, V, cout), device=dev, dtype=torch.bfloat16) if hs else new(B, Tp, V, cout)
        part = ops.rows_gemm(x if hs else x32(), W["res"], r, K=cin, N=cout, tmap=(1, s, 0, 0, 1), bias=P["residual.conv.bias"], stats=train)
        vec_r = _bn_vec(part, B * Tp * V, P, bufs, "residual.bn", train)
        o, o_sign = epilogue(u, vec_u, r, vec_r)
    if out_half and not pool_groups and o_sign is None:
        raise ops._lib.FgcnError("half-precision storage of the block's output needs the sign image (element count a multiple of 8)")
    S.update(u=u, vec_u=vec_u, r=r, vec_r=vec_r, o=None if pool_groups else o, o_sign=o_sign)
    return o, S


def pool_epilogue_ok(cfg: BlockConfig, B: int, T: int, V: int, groups: int) -> bool:
    """the last block's epilogue can carry the pooling: equal groups of whole samples, a sign image exists (element count and cout in 8s)"""
    Tp = (T - 1) // cfg.stride + 1
    return ops.paths().pool_epilogue and groups > 0 and B % groups == 0 and cfg.cout % 8 == 0 and (B * Tp * V * cfg.cout) % 8 == 0


# ---- backward --------------------------------------------------------------------------------------------------------
def zero_bias_floats(cfg: BlockConfig) -> int:
    """floats of exactly-zero bias gradients a block hands out in train mode (conv_d x 3, the temporal conv, down, residual conv)"""
    return (NUM_SUBSETS + 1 + int(cfg.has_down) + int(cfg.residual == "conv")) * cfg.cout


class _BiasGrads:
    """Gradients of conv biases that feed a BatchNorm.  In train mode the BatchNorm subtracts the batch mean, so the
    block output does not depend on such a bias and its gradient is exactly zero (the reference's autograd produces
    ~1e-9 rounding noise there): all of a block's are disjoint slices of ONE zero-filled
    allocation (one fill launch per block; every parameter still gets memory of its own).  Only eval-mode statistics make
    them real column sums."""

    def __init__(self, cfg: BlockConfig, device, train: bool, zeros: Optional[torch.Tensor] = None):
        self.train = train
        n = zero_bias_floats(cfg)
        # ``zeros``: this block's slice of a pool the MODEL filled once per step (one launch for the ten blocks instead of ten)
        if train and zeros is not None and zeros.numel() >= n:
            self.pool = zeros
        else:
            self.pool = torch.zeros(n, device=device, dtype=torch.float32) if train else None
        self.used = 0

    def __call__(self, d: torch.Tensor, c: int) -> torch.Tensor:
        if not self.train:
            return ops.col_sum(d, c)
        self.used += c
        return self.pool[self.used - c:self.used]


def block_backward(d_o: torch.Tensor, S: Dict[str, Optional[torch.Tensor]], P: Dict[str, torch.Tensor],
                   W: Dict[str, torch.Tensor], cfg: BlockConfig, train: bool = True, need_dx: bool = True,
                   pool: Optional[Tuple[int, tuple]] = None, zeros: Optional[torch.Tensor] = None):
    """-> (dx (B, T, V, cx) or None, {param name: grad in the parameter's own shape}).
    The leaf reductions of the block (weight-gradient slabs, adj_b, embedding-bias partials) are collected and issued as one
    launch at the end (ops.deferred_reductions).  Weight gradients are leaves of the backward graph and run in line, on the one stream:
    launching them on a second HIP stream beside the HBM-bound chain was built in round 2, lost its A/B in rounds 3 and 4 (the matrix
    kernels fill every CU's registers and LDS, so nothing of the other stream is co-resident: DESIGN.md section 3.2 item 11) and was
    removed in round 6."""
    with ops.deferred_reductions() as batch:
        out = _block_backward(d_o, S, P, W, cfg, train, need_dx, pool, zeros)
        batch.flush()
    return out


def _block_backward(d_o, S, P, W, cfg: BlockConfig, train: bool, need_dx: bool, pool=None, zeros=None):
    """``pool`` = (groups, shape of the block's output): the block's forward returned the per-group mean of its output (pool_groups) and
    ``d_o`` is the gradient of that, (groups, cout); it is consumed as a per-group row (divided by the group's rows) where the kernels
    take one (POOL_BACKWARD_ROWS) and expanded to the output's shape otherwise."""
    x = S["x"]
    B, T, V, cx = x.shape
    cout, ic, s = cfg.cout, cfg.ic, cfg.stride
    cin, cin_true = cx, cfg.cin  # kernels work on the padded channel count; gradients are cut back to cin_true
    Tp = d_o.shape[1] if pool is None else pool[1][1]
    dev = x.device
    o_ = ops.paths()             # this context's kernel-form options (the forward's context: ops.context_bound)
    half = bool(S.get("half")) and train     # G was stored as bfloat16: dU (the gradient of the temporal conv's output) is too
    new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
    G: Dict[str, torch.Tensor] = {}
    d_o = d_o.contiguous()
    o_numel = B * Tp * V * cout
    kt = P["tcn1.conv.weight"].shape[2]
    ha = half_activations_on(train, o_)
    x16 = x.dtype == torch.bfloat16          # the block's input arrived as bfloat16 (paths.half_activations): its gradient leaves as bfloat16
    # Identity shortcuts (cin == cout, stride 1: both the graph convolution's `y += x` and the block residual) send the ReLU-gated
    # incoming gradients straight to dx.  Instead of the BatchNorm-backward kernels writing / read-modify-writing dx, the kernel
    # that forms the spatial term of dx (joint_dagg) adds both from their sign images: two activation passes less per block.
    small = lambda width: B * max(T, Tp) * V * width * 4 < 0x7FFF0000      # noqa: E731  the tile kernels address with 32-bit byte offsets
    tile_ok = (o_.spatial_bwd_tile and cin >= o_.spatial_bwd_tile_min_cin and "d_t_b3" in W and ops.spatial_bwd_tile_available(V, cin, cout)
               and (ops.get_math_mode() in ("bf16x3", "bf16") or o_.spatial_bwd_tile_f16x2) and small(max(cin, cout)))
    gate_in_dagg = ((o_.gated_shortcuts_tile if tile_ok else o_.gated_shortcuts) and o_.fused_dagg and not cfg.has_down and cfg.residual == "identity"
                    and cx == cfg.cin and cout % 8 == 0 and S["o_sign"] is not None and S["g_sign"] is not None
                    and o_numel * 4 < 0x7FFF0000)
    # -- paths.half_activations: which gradients are bfloat16 tensors --------------------------------------------------------------------
    # dG (written by the temporal data gradient's halo kernel from a bfloat16 dU; not when that kernel's epilogue carries the BatchNorm sums)
    fuse_sums = (o_.get("bn_sums_in_dgrad", ops.get_math_mode()) and cout <= o_.bn_sums_max_c and train and s == 1 and not cfg.has_down and S["g_sign"] is not None
                 and "t_t4" in W and ops.tconv_halo_bn_sums())
    dg16 = bool(ha and half and kt > 1 and temporal_dgrad_records_amax(W, kt, s) and not fuse_sums and S["g_sign"] is not None)     # (narrowed below)
    wgrad_tile = (o_.spatial_wgrad_tile and x.shape[3] == cin and ops.spatial_wgrad_tile_available(V, cin, cout) and small(max(cin, cout))
                  and (ops.get_math_mode() in ("bf16x3", "bf16") or o_.spatial_wgrad_tile_f16x2))
    half_dy = bool(train and ops.get_math_mode() == "bf16" and o_.get("half_storage", "bf16") and wgrad_tile and tile_ok and x.shape[3] == cin
                   and S["g_sign"] is not None)
    emb_tile_bwd = (not cfg.static_adjacency) and emb_bwd_tile_ok(W, cfg, B, T, V, cx, o_) and x.shape[3] == cin
    # dx itself: every writer of dx must have the bfloat16 form -- the fused spatial backward first (with both gated shortcuts, or none to
    # add), then the embedding tile kernel; a residual / down conv or an ungated shortcut writes float32, and the block converts at the end
    dx16 = bool(x16 and ha and tile_ok and x.shape[3] == cin and half_dy and not cfg.has_down and cfg.residual != "conv"
                and (cfg.residual == "none" or gate_in_dagg)
                and (cfg.static_adjacency or (emb_tile_bwd and S["emb"] is not None and S["emb"].dtype == torch.bfloat16))
                and (not gate_in_dagg or (dg16 and (d_o.dtype == torch.bfloat16 or (pool is not None and o_.pool_backward_rows)))))
    dg16 = dg16 and (dx16 or not gate_in_dagg)       # (a gated addend has dx's storage type: the fused backward adds it)
    x_h = x                                  # the input as it arrived: the kernels with a typed form take the bfloat16 x beside a bfloat16 dy / emb / gradient
    # the shortcut branches' gradients (dd / dr) as bfloat16 and their kernels on the bfloat16 x: the typed row GEMMs / 1x1 weight gradient
    hs = bool(x16 and ha and o_.half_shortcuts and cin % 32 == 0 and cout % 8 == 0)
    _xf: List[torch.Tensor] = []

    def xf() -> torch.Tensor:                # x for a kernel without a bfloat16-input form (converted once, on first use)
        if not x16:
            return x_h
        if not _xf:
            _xf.append(x_h.float())
        return _xf[0]
    del x                                    # (every use below names the form it needs: x_h, or xf())
    dx = torch.empty((B, T, V, cx), device=dev, dtype=torch.bfloat16) if dx16 else new(B, T, V, cx)
    dx_live = False      # becomes True once dx holds a valid partial sum
    as_extra = lambda t: t if (dx16 or t.dtype == torch.float32) else t.float()      # noqa: E731  a gated addend has dx's storage type
    gated: List[tuple] = []
    grp_rows = grp_samples = 0
    if pool is not None:
        groups = pool[0]
        rows = o_numel // cout // groups
        d_o = d_o / rows                             # (groups, cout): every row of a group receives the group's gradient / rows
        if o_.pool_backward_rows and (not gate_in_dagg or (tile_ok and x_h.shape[3] == cin)):
            grp_rows, grp_samples = rows, B // groups        # the BatchNorm-backward passes and the gated addend read the group's row
        else:
            d_o = d_o.unsqueeze(1).expand(groups, rows, cout).contiguous().view(B, Tp, V, cout)
    bias_grad = _BiasGrads(cfg, dev, train, zeros)

    # -- O = relu(BN(u) + res) ---------------------------------------------------------------------------------------------
    if cfg.residual == "none":
        du, _, sums = ops.bn_act_bwd(d_o, S["o"], S["u"], S["vec_u"], None, None, res_mode=0, train=train,
                                     sign_mask=S["o_sign"], grp_rows=grp_rows, da_bf16=half)
    elif cfg.residual == "identity" and gate_in_dagg:
        du, _, sums = ops.bn_act_bwd(d_o, S["o"], S["u"], S["vec_u"], x_h, None, res_mode=1, train=train, need_db=False,
                                     sign_mask=S["o_sign"], grp_rows=grp_rows, da_bf16=half)
        gated.append((d_o, S["o_sign"], grp_samples) if grp_samples else (as_extra(d_o), S["o_sign"]))   # dx += d_o * [o > 0], added by joint_dagg below
    elif cfg.residual == "identity":
        du, _, sums = ops.bn_act_bwd(d_o, S["o"], S["u"], S["vec_u"], x_h, None, res_mode=1, train=train, db=dx,
                                     sign_mask=S["o_sign"], grp_rows=grp_rows, da_bf16=half)
        dx_live = True
    else:
        du, dr, sums = ops.bn_act_bwd(d_o, S["o"], S["u"], S["vec_u"], S["r"], S["vec_r"], res_mode=2, train=train,
                                      sign_mask=S["o_sign"], grp_rows=grp_rows, da_bf16=half, db_bf16=hs)
        G["residual.bn.weight"], G["residual.bn.bias"] = sums[2], sums[0].clone()   # own memory: sums[0] is tcn1.bn.bias too
        ops.rows_gemm(dr, W["res_t"], dx, K=cout, N=cx, tmap=(1, 1, 0, 0, s))   # frames t % s != 0 receive zeros
        dx_live = True
        G["residual.conv.weight"] = ops.rows_wgrad(x_h if dr.dtype == torch.bfloat16 else xf(), dr, K=cin, N=cout, tmap=(1, s, 0, 0, 1),
                                                   conv_param=(1, cin_true))
        G["residual.conv.bias"] = bias_grad(dr, cout)
    G["tcn1.bn.weight"], G["tcn1.bn.bias"] = sums[1], sums[0]

    # -- temporal conv -------------------------------------------------------------------------------------------------------
    dg = torch.empty((B, T, V, cout), device=dev, dtype=torch.bfloat16) if dg16 else new(B, T, V, cout)
    # identity blocks in the split-bf16 modes: the data-gradient kernel sums dg * [g > 0] and dg * [g > 0] * y_hat in its epilogue,
    # so the BatchNorm backward of the graph convolution below needs no reduction pass of its own over dg and y (fuse_sums, above)
    # math mode f16x2: the data-gradient kernels record the largest magnitudes of the tensors they stage (slot 0 = du, 1 = demb);
    # with the forward's slots they are the operand scales of the weight gradients, which therefore follow those kernels
    f16x2 = S.get("amax") is not None and ops.get_math_mode() == "f16x2"
    bamax = torch.zeros(4, device=dev, dtype=torch.int32) if f16x2 else None
    # (a slot only counts when the kernel that ran really recorded it: the row-GEMM fallbacks of odd paddings / T == 1 record
    # nothing, and a weight gradient scaled by a zero-initialised slot would silently leave the f16 range)
    du_amax = f16x2 and temporal_dgrad_records_amax(W, kt, s)
    g_partials = temporal_dgrad(du, dg, W, kt, s, bn_bwd=(S["y"], S["g_sign"], S["vec_y"]) if fuse_sums else None,
                                amax_out=bamax[0:1] if du_amax else None)
    # weight gradients are reduced straight into the parameter's (out, in, kt, 1) layout: autograd takes them as they are
    G["tcn1.conv.weight"] = ops.tconv_wgrad(S["g"], du, taps=kt, stride=s, conv_param=(1, cout),
                                            amax=(S["amax"][1:2], bamax[0:1]) if du_amax and S.get("g_amax") else None)
    G["tcn1.conv.bias"] = bias_grad(du, cout)

    # -- G = relu(BN(y) + down(x)) ---------------------------------------------------------------------------------------------
    # math mode bf16, training, both consumers of dy on their tile kernels: dy is stored as bfloat16 (only their staging reads it;
    # wgrad_tile / half_dy: above)
    if cfg.has_down:
        dy, dd, sums = ops.bn_act_bwd(dg, S["g"], S["y"], S["vec_y"], S["d"], S["vec_d"], res_mode=2, train=train,
                                      sign_mask=S["g_sign"], da_bf16=half_dy, db_bf16=hs)
        G["gcn1.down.1.weight"], G["gcn1.down.1.bias"] = sums[2], sums[0].clone()   # own memory: sums[0] is gcn1.bn.bias too
        pw_gemm(dd, W, "down_t", dx, K=cout, N=cx, accumulate=dx_live)
        dx_live = True
        G["gcn1.down.0.weight"] = ops.rows_wgrad(x_h if dd.dtype == torch.bfloat16 else xf(), dd, K=cin, N=cout, conv_param=(1, cin_true))
        G["gcn1.down.0.bias"] = bias_grad(dd, cout)
    elif gate_in_dagg:
        dy, _, sums = ops.bn_act_bwd(dg, S["g"], S["y"], S["vec_y"], x_h, None, res_mode=1, train=train, need_db=False,
                                     sign_mask=S["g_sign"], partials=g_partials if fuse_sums else None, da_bf16=half_dy)
        gated.append((as_extra(dg), S["g_sign"]))  # dx += dg * [g > 0]
    else:
        dy, _, sums = ops.bn_act_bwd(dg, S["g"], S["y"], S["vec_y"], x_h, None, res_mode=1, train=train, db=dx,
                                     db_accumulate=dx_live, sign_mask=S["g_sign"], partials=g_partials if fuse_sums else None, da_bf16=half_dy)
        dx_live = True
    G["gcn1.bn.weight"], G["gcn1.bn.bias"] = sums[1], sums[0]

    # -- conv_d and the joint aggregation ------------------------------------------------------------------------------------------
    a_hat = S["a_hat"]
    c3 = 3 * cin
    bwd_tile = tile_ok and x_h.shape[3] == cin and (not gated or len(gated) == 2)
    if half_dy and not bwd_tile:     # (cannot happen: the gated list holds none or both shortcuts by construction)
        raise ops._lib.FgcnError("block backward: dy was stored as bfloat16 but the fused spatial backward is not taken")
    dagg, dy_amax = None, False
    if not bwd_tile:
        # dagg = dy . Wd first: in math mode f16x2 the row GEMM records max |dy| (slot 3), the operand scale of conv_d's weight gradient
        dagg = new(B, T, V, c3)
        dy_amax = f16x2 and pw_routed(W, "d_t", dy, cout)
        pw_gemm(dy, W, "d_t", dagg, K=cout, N=c3, amax_out=bamax[3:4] if dy_amax else None)
    # weight gradient of conv_d: agg is recomputed (cheaper than keeping 3 activations per block) and contracted with dy
    if wgrad_tile:
        gw = ops.spatial_wgrad_tile(x_h if dy.dtype == torch.bfloat16 else xf(), dy, a_hat, conv_param=(NUM_SUBSETS, cin_true))   # agg on chip, whole frame tiles
    elif o_.fused_agg_wgrad and x_h.shape[3] == cin and cin >= 32 and cout <= o_.get("fused_agg_wgrad_max_cout", ops.get_math_mode()):
        # agg = x . A^ is formed in registers and contracted with dy at once: never written
        gw = ops.spatial_wgrad(xf(), dy, a_hat, conv_param=(NUM_SUBSETS, cin_true))
    else:
        agg = new(B, T, V, c3)
        agg_amax = mix_agg(xf(), agg, a_hat, cin, amax_out=bamax[2:3] if dy_amax else None)
        gw = ops.rows_wgrad(agg, dy, K=3 * cin, N=cout, conv_param=(NUM_SUBSETS, cin_true),   # (3, cout, cin_true, 1, 1)
                            amax=(bamax[2:3], bamax[3:4]) if agg_amax else None)
        del agg
    dbias = None if train else ops.col_sum(dy, cout)
    for k in range(NUM_SUBSETS):
        G[f"gcn1.conv_d.{k}.weight"] = gw[k]
        # three parameters, three buffers (the sum of the three biases is what the kernel adds: equal gradients)
        G[f"gcn1.conv_d.{k}.bias"] = bias_grad(dy, cout) if train else (dbias if k == 0 else dbias.clone())
    if bwd_tile:
        # dagg on chip: dx and dA^ in one launch (the bfloat16 x beside a bfloat16 dy, whatever dx is)
        part = ops.spatial_bwd_tile(dy, x_h if dy.dtype == torch.bfloat16 else xf(), a_hat, W["d_t_b3"], dx, accumulate=dx_live, gated=gated)
    elif o_.fused_dagg and x_h.shape[3] == cin:
        part = ops.joint_dagg(xf(), dagg, a_hat, dx, accumulate=dx_live, gated=gated)   # dx and dA^ from one pass over dagg
    else:
        mix_dx(dagg, dx, a_hat, cin, accumulate=dx_live)
        part = ops.joint_gram(xf(), dagg, [(0, k * cin, cin) for k in range(NUM_SUBSETS)])
    dx_live = True
    d_a_hat, d_s = ops.adj_softmax_bwd(part, 1.0 / (ic * T), S["c_mat"], V)
    db = torch.empty_like(P["gcn1.adj_b"])
    ops.reduce_sum(d_a_hat.view(B, -1), db.view(-1), leaf=True)
    G["gcn1.adj_b"] = db

    # -- attention embeddings -----------------------------------------------------------------------------------------------------
    if not cfg.static_adjacency:
        emb = S["emb"]
        if emb_tile_bwd:
            # demb on chip: dx += demb . Wemb, then (a leaf) dWemb = demb^T . x and the bias gradient
            if x16 and dx.dtype == torch.float32 and emb.dtype == torch.bfloat16 and dx_live:
                # the last writer of this block's float32-accumulated dx hands it over as the bfloat16 tensor the bfloat16 input asks for
                dx16_out = torch.empty(dx.shape, device=dev, dtype=torch.bfloat16)
                ops.emb_dx_tile(emb, d_s, W["emb_t_b3"], dx16_out, ic=ic, accumulate=True, dx_old=dx)
                dx = dx16_out
            else:
                ops.emb_dx_tile(emb, d_s, W["emb_t_b3"], dx, ic=ic, accumulate=dx_live)
            gw, gb = ops.emb_wgrad_tile(emb, x_h if emb.dtype == torch.bfloat16 else xf(), d_s, ic=ic)
            gw = gw.view(6 * ic, cin_true, 1, 1)
        else:
            if emb.dtype != torch.float32:       # (cannot happen: the forward asked the same predicate before it chose the storage)
                raise ops._lib.FgcnError("block backward: emb was stored as bfloat16 but the tile form of its backward is not taken")
            demb = new(B, T, V, 6 * ic)
            gb = mix_demb(emb, demb, d_s, ic)                                             # + column sums = bias gradient
            demb_amax = f16x2 and S["x_amax"] and pw_routed(W, "emb_t", demb, 6 * ic)
            pw_gemm(demb, W, "emb_t", dx, K=6 * ic, N=cx, accumulate=dx_live, amax_out=bamax[1:2] if demb_amax else None)
            gw = ops.rows_wgrad(xf(), demb, K=cin, N=6 * ic, conv_param=(1, cin_true),     # (6ic, cin_true, 1, 1)
                                amax=(S["amax"][0:1], bamax[1:2]) if demb_amax else None)
        for k in range(NUM_SUBSETS):
            for j, grp in enumerate(("conv_a", "conv_b")):
                lo = (2 * k + j) * ic
                G[f"gcn1.{grp}.{k}.weight"] = gw[lo:lo + ic]
                G[f"gcn1.{grp}.{k}.bias"] = gb[lo:lo + ic]
    if x16 and dx.dtype != torch.bfloat16:
        dx = dx.to(torch.bfloat16)           # the gradient of a bfloat16 input (one rounding, as the bfloat16-writing kernels apply it)
    return (dx if need_dx else None), G


# ---- autograd ----------------------------------------------------------------------------------------------------------
class STBlockFunction(torch.autograd.Function):
    """y = SpatialTemporalConv(x); inputs: x, then the block's parameters in ``param_names(cfg)`` order."""

    @staticmethod
    def forward(ctx, x, cfg: BlockConfig, train: bool, bufs: Dict[str, torch.Tensor], W: Dict[str, torch.Tensor],
                holder: Optional[dict], *params):
        names = param_names(cfg)
        P = dict(zip(names, params))
        pool_groups = holder.get("pool_groups", 0) if holder is not None else 0
        inference = bool(holder.get("inference")) if holder is not None else False      # eval mode and no autograd graph (the module says)
        out_half = bool(holder.get("out_half")) if holder is not None else False        # the consumer takes a bfloat16 output (paths.half_activations)
        o, S = block_forward(x, P, bufs, W, cfg, train, pool_groups, inference=inference, out_half=out_half)
        B, T, V, _ = x.shape
        ctx.pool = (pool_groups, (B, (T - 1) // cfg.stride + 1, V, cfg.cout)) if pool_groups else None
        ctx.zeros = holder.get("zeros") if holder is not None else None      # the block's slice of the model's zero pool (or None)
        # the block's input and output go through save_for_backward (an output kept on ctx would be a reference cycle
        # o -> grad_fn -> ctx -> o that only the garbage collector frees); the other activations are private to the block
        # (the backward gates on the one-bit sign image of o, so o itself is only kept when that image does not exist;
        # nn.Dropout(inplace=True) after the block may then overwrite o freely, as in the reference's Model)
        keep_o = S["o_sign"] is None
        S["x"] = S["o"] = None
        ctx.cfg, ctx.train, ctx.names, ctx.W, ctx.S, ctx.keep_o = cfg, train, names, W, S, keep_o
        ctx.save_for_backward(x, *((o,) if keep_o else ()), *params)
        if holder is not None:
            holder["adj_c"] = S["c_mat"]
        return o

    @staticmethod
    def backward(ctx, d_o):
        if ctx.S is None:
            raise RuntimeError("STBlockFunction: backward ran twice -- the block frees its saved activations after the first "
                               "backward (retain_graph is not supported; run the forward again)")
        x, *params = ctx.saved_tensors
        o = params.pop(0) if ctx.keep_o else None
        P = dict(zip(ctx.names, params))
        S = dict(ctx.S, x=x, o=o)
        ctx.S = None
        dx, G = block_backward(d_o, S, P, ctx.W, ctx.cfg, ctx.train, need_dx=ctx.needs_input_grad[0], pool=ctx.pool, zeros=ctx.zeros)
        del S
        grads = []
        for i, n in enumerate(ctx.names):
            g = G.get(n) if ctx.needs_input_grad[6 + i] else None
            if g is None and ctx.needs_input_grad[6 + i]:
                g = torch.zeros_like(P[n])          # static-adjacency: embedding convs get no gradient
            grads.append(g)
        return (dx, None, None, None, None, None, *grads)


class LinearFunction(torch.autograd.Function):
    """logits = h . W^T + b on the row GEMM (reference: nn.Linear `fc`, mmargcn/agcn.py:177,200): keeps the classifier off
    hipBLASLt, whose `UserArgs` kernels do not survive HIP-graph replay at small row counts (the 8-clip shard of a
    strong-scaling run produced wrong logits from the second replay on)."""

    @staticmethod
    def forward(ctx, h: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        n, k = h.shape
        classes = weight.shape[0]
        npad = _r4(classes)
        wt = torch.zeros((1, k, npad), device=h.device, dtype=torch.float32)
        wt[0, :, :classes] = weight.t()
        bp = None
        if bias is not None:
            bp = torch.zeros(npad, device=h.device, dtype=torch.float32)
            bp[:classes] = bias
        hc = h.contiguous()
        out = torch.empty((n, 1, 1, npad), device=h.device, dtype=torch.float32)
        ops.rows_gemm(hc.view(n, 1, 1, k), wt, out, K=k, N=npad, bias=bp)
        ctx.save_for_backward(hc, weight)
        ctx.has_bias = bias is not None
        return out.view(n, npad)[:, :classes]

    @staticmethod
    def backward(ctx, d_out: torch.Tensor):
        hc, weight = ctx.saved_tensors
        n, k = hc.shape
        classes = weight.shape[0]
        npad = _r4(classes)
        dl = torch.zeros((n, 1, 1, npad), device=hc.device, dtype=torch.float32)
        dl.view(n, npad)[:, :classes] = d_out
        dh = dw = db = None
        if ctx.needs_input_grad[0]:
            w = torch.zeros((1, npad, k), device=hc.device, dtype=torch.float32)
            w[0, :classes] = weight
            dh = torch.empty((n, 1, 1, k), device=hc.device, dtype=torch.float32)
            ops.rows_gemm(dl, w, dh, K=npad, N=k)
            dh = dh.view(n, k)
        if ctx.needs_input_grad[1]:
            gw = ops.rows_wgrad(hc.view(n, 1, 1, k), dl, K=k, N=npad, wide=False)      # (1, k, npad)
            dw = gw[0, :, :classes].t().contiguous()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.col_sum(dl.view(n, 1, 1, npad), npad)[:classes].contiguous()
        return dh, dw, db


class GroupMeanFunction(torch.autograd.Function):
    """(G, R, C) -> (G, C) global average pooling on fgcn_group_mean (reference: x.view(N, M, c, -1).mean(3).mean(1),
    mmargcn/agcn.py:196-197 -- equal-sized groups, so one mean over all M*T'*V rows of a clip)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor):
        ctx.shape = tuple(x.shape)
        return ops.group_mean(x.contiguous())

    @staticmethod
    def backward(ctx, d_out: torch.Tensor):
        G, R, C = ctx.shape
        return (d_out / R).unsqueeze(1).expand(G, R, C).contiguous()


class DataBNFunction(torch.autograd.Function):
    """data_bn (nn.BatchNorm1d over the (m, v, c) channels of the network input, statistics across (n, t); reference
    mmargcn/agcn.py:150,186-188) fused with the layout change the blocks need: (N, M, T, V, C) -> (N*M, T, V, C padded to 4).
    The nn.BatchNorm1d module is the parameter / buffer container (momentum 0.1-style exponential running statistics)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, train: bool, momentum: float, eps: float):
        x = x.contiguous()
        N, M, T, V, C = x.shape
        if train:
            vec = ops.bn_finalize(ops.data_bn_stats(x), N * T, weight, bias, running_mean, running_var, momentum, eps)
        else:
            vec = ops.bn_eval_coeffs(weight, bias, running_mean, running_var, eps)
        ctx.save_for_backward(x, vec)
        ctx.train = train
        return ops.data_bn_apply(x, vec, _r4(C))

    @staticmethod
    def backward(ctx, d_out):
        x, vec = ctx.saved_tensors
        dgamma, dbeta, dx = ops.data_bn_bwd(d_out.contiguous(), x, vec, ctx.train, ctx.needs_input_grad[0])
        return dx, dgamma, dbeta, None, None, None, None, None


def data_bn(x: torch.Tensor, bn: torch.nn.BatchNorm1d) -> torch.Tensor:
    """The model's input stage on libfgcn: ``bn`` = the model's ``data_bn`` module; updates its running statistics and batch
    counter in train mode like the module's own forward would."""
    if bn.momentum is None or not bn.affine or not bn.track_running_stats:
        raise NotImplementedError("data_bn: affine BatchNorm1d with exponential running statistics (the reference's default) only")
    if x.shape[1] * x.shape[3] * x.shape[4] != bn.num_features:
        raise ValueError(f"data_bn: input {tuple(x.shape)} does not have {bn.num_features} (m, v, c) channels")
    out = DataBNFunction.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps)
    if bn.training:
        bn.num_batches_tracked.add_(1)
    return out


class CrossEntropyFunction(torch.autograd.Function):
    """nn.CrossEntropyLoss() / F.cross_entropy(logits, labels), mean reduction (reference session/session.py:53): one
    fixed-order kernel each way."""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, probs = ops.cross_entropy_fwd(logits, labels)
        ctx.save_for_backward(probs, labels, loss)
        return loss[0]

    @staticmethod
    def backward(ctx, d_loss):
        probs, labels, loss = ctx.saved_tensors
        return ops.cross_entropy_bwd(probs, labels, loss, d_loss.contiguous().view(1)), None


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    return CrossEntropyFunction.apply(logits, labels)


ops.bind_all_functions(globals())     # every Function's backward runs in its forward's library context (ops.Context)
