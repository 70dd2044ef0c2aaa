"""1-D graph convolutions for IMU graphs (reference torch_src/models/mmargcn/graph_convolution.py).

``STGCNGraphConvolution`` (:12-52): ``relu(Conv1d_1x1(x) . adj^T + residual(x))`` on ``x (B, F, V)`` with one static V x V
adjacency, V = sequence_length * num_signals nodes (up to ~2000).  On the MI355X the layer is three existing libfgcn entry
points plus a transpose (include/fgcn.h):

  * activations travel between layers node-major ``(B, V, F)`` (channels-last, the layout of every other kernel here);
  * the 1x1 Conv1d is the row GEMM over the ``B*V`` node rows (``fgcn_rows_gemm`` / the split-bf16 1x1 kernel);
  * ``torch.matmul(support, adj.t())`` contracts over the nodes: the support is transposed to feature-major ``(B, O, Vp)``
    (``fgcn_transpose``, Vp = V padded to a multiple of 64 with zero columns) and multiplied with the adjacency as a SHARED
    (Vp x Vp) weight by the same row GEMM (``B*O`` rows), then transposed back;
  * residual (identity, or Conv1d + BatchNorm1d with batch statistics from the GEMM epilogue) + ReLU = ``fgcn_bn_act``.

``AGCNGraphConvolution`` (:56-113: per-sample V x V attention) reuses the same pieces with per-sample matrices as the GEMM
weight and a row softmax on the transposed scores (``fgcn_row_softmax_*``); see ``_AgcnConv1dFunction``.
"""
from __future__ import annotations

from typing import Dict

import weakref

import torch
import torch.nn as nn

from ... import ops
from ...block import pw_gemm


def _r(c: int, m: int) -> int:
    return (c + m - 1) // m * m


_SPLIT_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _split_form(mod, name: str, w: torch.Tensor):
    """The split form of a (1, K, N) matrix that is rebuilt every call (a transposed / padded weight) in the products of the current math
    mode.  bf16x3 / bf16: one ``fgcn_pack_split3`` launch.  f16x2: the FGCN_PACK_SPLIT2H form goes through a pack plan whose device-side
    table is uploaded when it is built -- not allowed inside a HIP-graph capture -- so the module keeps one staging buffer + plan per
    name (built by the eager warm-up steps); a call copies into the staging buffer and runs the plan: two launches, capture-safe."""
    if ops.get_math_mode() != "f16x2":
        return ops.pack_split3(w)
    from ...packing import Form, PackPlan, Seg
    cache = _SPLIT_CACHE.setdefault(mod, {})          # (kept beside the module, not in it: state_dict / pickling never see it)
    ent = cache.get(name)
    if ent is None or ent[0].shape != w.shape or ent[0].device != w.device:
        stage = torch.empty_like(w)
        _, K, N = w.shape
        f = Form("split2h", 1, K, N, [Seg(stage, st_tap=K * N, st_k=N, st_n=1, klen=K, nlen=N, tlen=1)])
        f.alloc(w.device)
        ent = cache[name] = (stage, f, PackPlan([f]))
    ent[0].copy_(w)
    ent[2].run()
    return ent[1].dst


def _rows4(t: torch.Tensor) -> torch.Tensor:
    """(B, R, C) -> the (B, R, 1, C) view the row kernels index as (sample, frame, joint, channel)."""
    return t.view(t.shape[0], t.shape[1], 1, t.shape[2])


def _identity_vec(c: int, device) -> torch.Tensor:
    """(4, C) = {mean 0, rstd 1, scale 1, shift 0}: fgcn_bn_act's coefficient vector of a tensor that has no BatchNorm."""
    v = torch.zeros((4, c), device=device, dtype=torch.float32)
    v[1:3] = 1.0
    return v


class _GraphConv1dFunction(torch.autograd.Function):
    """x_nm (B, V, Fp) -> relu(conv(x) . adj^T + residual(x)) (B, V, O), all arithmetic in libfgcn kernels."""

    @staticmethod
    def forward(ctx, x, mod: "STGCNGraphConvolution", train: bool, weight, bias, res_w, res_b, res_g, res_beta):
        B, V, Fp = x.shape
        O, Fin = weight.shape[0], weight.shape[1]
        dev = x.device
        A = mod._adjacency_forms()
        Vp = A["Vp"]
        with torch.no_grad():
            w = torch.zeros((1, Fp, O), device=dev, dtype=torch.float32)
            w[0, :Fin] = weight.view(O, Fin).t()
        W: Dict[str, torch.Tensor] = {"w": w}
        if ops.get_math_mode() in ops.SPLIT_MODES and Fp % 64 == 0:
            W["w_s3"] = _split_form(mod, "w", w)        # split form of the current products (bf16x3 / f16x2)
        support = torch.empty((B, V, O), device=dev, dtype=torch.float32)
        pw_gemm(_rows4(x), W, "w", _rows4(support), K=Fp, N=O, bias=bias)
        sup_fm = ops.transpose(support, Vp)                                   # (B, O, Vp), zero padding columns
        out_fm = torch.empty((B, O, Vp), device=dev, dtype=torch.float32)
        pw_gemm(_rows4(sup_fm), A, "adjT", _rows4(out_fm), K=Vp, N=Vp)
        main = ops.transpose_into(out_fm, V)                                  # (B, V, O)
        vec_id = _identity_vec(O, dev)
        r = vec_r = None
        if mod.res_kind == "none":
            out, mask = ops.bn_act(main, vec_id, None, None, relu=True, sign_mask=True)
        elif mod.res_kind == "identity":
            out, mask = ops.bn_act(main, vec_id, x, None, relu=True, sign_mask=True)
        else:
            with torch.no_grad():
                wr = torch.zeros((1, Fp, O), device=dev, dtype=torch.float32)
                wr[0, :Fin] = res_w.view(O, Fin).t()
            W["wr"] = wr
            if "w_s3" in W:
                W["wr_s3"] = _split_form(mod, "wr", wr)
            r = torch.empty((B, V, O), device=dev, dtype=torch.float32)
            part = pw_gemm(_rows4(x), W, "wr", _rows4(r), K=Fp, N=O, bias=res_b, stats=train)
            bn = mod.residual[1]
            vec_r = (ops.bn_finalize(part, B * V, res_g, res_beta, bn.running_mean, bn.running_var) if train
                     else ops.bn_eval_coeffs(res_g, res_beta, bn.running_mean, bn.running_var))
            if train:
                bn.num_batches_tracked += 1
            out, mask = ops.bn_act(r, vec_r, main, None, relu=True, sign_mask=True)   # relu(BN(r) + main)
        ctx.mod, ctx.train, ctx.W = mod, train, W
        ctx.save_for_backward(x, out, mask, r, vec_r, weight, res_w)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, out, mask, r, vec_r, weight, res_w = ctx.saved_tensors
        mod, train, W = ctx.mod, ctx.train, ctx.W
        B, V, Fp = x.shape
        O, Fin = weight.shape[0], weight.shape[1]
        dev = x.device
        A = mod._adjacency_forms()
        Vp = A["Vp"]
        d_out = d_out.contiguous()
        vec_id = _identity_vec(O, dev)
        dx = None
        g_res_w = g_res_b = g_res_g = g_res_beta = None
        if mod.res_kind == "none":
            d_main, _, _ = ops.bn_act_bwd(d_out, out, out, vec_id, None, None, res_mode=0, train=False, sign_mask=mask, need_sums=False)
        elif mod.res_kind == "identity":
            dx = torch.empty_like(x)
            d_main, _, _ = ops.bn_act_bwd(d_out, out, out, vec_id, x, None, res_mode=1, train=False, db=dx, sign_mask=mask,
                                          need_sums=False)
        else:
            d_main = torch.empty((B, V, O), device=dev, dtype=torch.float32)
            dr, _, sums = ops.bn_act_bwd(d_out, out, r, vec_r, out, None, res_mode=1, train=train, db=d_main, sign_mask=mask)
            g_res_g, g_res_beta = sums[1], sums[0]
            g_res_w = ops.rows_wgrad(_rows4(x), _rows4(dr), K=Fp, N=O, conv_param=(1, Fin)).view(O, Fin, 1)
            g_res_b = torch.zeros(O, device=dev, dtype=torch.float32) if train else ops.col_sum(_rows4(dr), O)
            dx = torch.empty_like(x)
            Wt = {"wr_t": W["wr"][0].t().contiguous().unsqueeze(0)}           # (1, O, Fp)
            if "wr_s3" in W and O % 32 == 0:
                Wt["wr_t_s3"] = _split_form(mod, "wr_t", Wt["wr_t"])
            pw_gemm(_rows4(dr), Wt, "wr_t", _rows4(dx), K=O, N=Fp)
        # main path: d_support = d_main . adj  (feature-major), then the conv's data and weight gradients
        dm_fm = ops.transpose(d_main, Vp)
        ds_fm = torch.empty((B, O, Vp), device=dev, dtype=torch.float32)
        pw_gemm(_rows4(dm_fm), A, "adj", _rows4(ds_fm), K=Vp, N=Vp)
        d_support = ops.transpose_into(ds_fm, V)
        g_w = ops.rows_wgrad(_rows4(x), _rows4(d_support), K=Fp, N=O, conv_param=(1, Fin)).view(O, Fin, 1)
        g_b = ops.col_sum(_rows4(d_support), O)
        if ctx.needs_input_grad[0]:
            Wt = {"w_t": W["w"][0].t().contiguous().unsqueeze(0)}              # (1, O, Fp)
            if "w_s3" in W and O % 32 == 0:
                Wt["w_t_s3"] = _split_form(mod, "w_t", Wt["w_t"])
            acc = dx is not None
            if dx is None:
                dx = torch.empty_like(x)
            pw_gemm(_rows4(d_support), Wt, "w_t", _rows4(dx), K=O, N=Fp, accumulate=acc)
        else:
            dx = None
        return dx, None, None, g_w, g_b, g_res_w, g_res_b, g_res_g, g_res_beta


class STGCNGraphConvolution(nn.Module):
    """Same constructor, parameters and state-dict keys as the reference class; ``forward`` takes and returns the node-major
    image ``(B, V, Fp)`` (``Fp`` = in_features rounded up to 4, extra channels zero) -- ``GCN`` converts at its boundary."""

    def __init__(self, in_features: int, out_features: int, adj: torch.Tensor, bias: bool = True, residual: bool = True,
                 **kwargs):
        super().__init__()
        dropout = kwargs.get("dropout", 0.)
        self.sparse = kwargs.get("sparse", False)      # the reference's sparse path computes the same product
        if out_features % 4:
            raise ValueError(f"HIP graph convolution needs out_features % 4 == 0 (got {out_features})")
        if not bias:
            raise NotImplementedError("bias=False is not built (the reference never uses it)")
        self.in_features, self.out_features = in_features, out_features
        self.conv = nn.Conv1d(in_features, out_features, 1, bias=bias)
        if adj.is_sparse:
            adj = adj.to_dense()
        self.register_buffer("adj", adj.to(torch.float32))
        self.dropout = nn.Dropout(dropout) if dropout > 0 else None
        if dropout > 0:
            raise NotImplementedError("dropout inside the fused graph convolution is not built (reference default: 0)")
        if not residual:
            self.res_kind, self.residual = "none", None
        elif in_features == out_features:
            self.res_kind, self.residual = "identity", None
        else:
            self.res_kind = "conv"
            self.residual = nn.Sequential(nn.Conv1d(in_features, out_features, 1), nn.BatchNorm1d(out_features))
        self._adj_cache = None

    def _adjacency_forms(self) -> Dict[str, object]:
        """adj^T (forward) and adj (data gradient) as shared (1, Vp, Vp) row-GEMM weights, Vp = V padded to 64 with zeros, and
        their split forms in the bf16 math modes; cached per (device, math mode)."""
        key = (self.adj.device, self.adj.data_ptr(), ops.get_math_mode())
        if self._adj_cache is None or self._adj_cache[0] != key:
            V = self.adj.shape[0]
            Vp = _r(V, 64)
            with torch.no_grad():
                a = torch.zeros((Vp, Vp), device=self.adj.device, dtype=torch.float32)
                a[:V, :V] = self.adj
                forms = {"Vp": Vp, "adjT": a.t().contiguous().unsqueeze(0), "adj": a.contiguous().unsqueeze(0)}
                if ops.get_math_mode() in ops.SPLIT_MODES:
                    forms["adjT_s3"] = ops.pack_conv(forms["adjT"])
                    forms["adj_s3"] = ops.pack_conv(forms["adj"])
            self._adj_cache = (key, forms)
        return self._adj_cache[1]

    def recording_pins(self) -> list:
        """GraphStep hook: the padded adjacency forms a recording made now reads."""
        return [] if self._adj_cache is None else [self._adj_cache[1]]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        res = self.residual
        return _GraphConv1dFunction.apply(
            x, self, self.training, self.conv.weight, self.conv.bias,
            res[0].weight if res is not None else None, res[0].bias if res is not None else None,
            res[1].weight if res is not None else None, res[1].bias if res is not None else None)


class _AgcnConv1dFunction(torch.autograd.Function):
    """AGCNGraphConvolution.forward (graph_convolution.py:91-113) on node-major x (B, V, Fp), V % 4 == 0.

    Per-sample V x V products are row GEMMs with one sample's matrix as the weight, B problems per launch
    (``fgcn_rows_gemm_batched``).  The scores are formed TRANSPOSED,
    S^T_k[b][w][v] = phi_k[w] . theta_k[v] / ic, so the reference's softmax over dim -2 runs along the contiguous axis
    (fgcn_row_softmax_*) and A^^T_k = C^T_k + (adj_a + adj_b)_k^T is directly the row operand of the aggregation
    agg_k[b] = A^^T_k[b] . x[b]."""

    @staticmethod
    def forward(ctx, x, mod, train, adj_b, *params):
        B, V, Fp = x.shape
        O, ic, Fin = mod.out_features, mod.inter_c, mod.in_features
        dev = x.device
        new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)      # noqa: E731
        P = dict(zip(mod._param_names(), params))
        with torch.no_grad():
            rows = []
            for k in range(3):
                rows += [P[f"conv_a.{k}.weight"].view(ic, Fin), P[f"conv_b.{k}.weight"].view(ic, Fin)]
            w_emb = torch.zeros((1, Fp, 6 * ic), device=dev, dtype=torch.float32)
            w_emb[0, :Fin] = torch.cat(rows, 0).t()                                   # [th0 ph0 th1 ph1 th2 ph2]
            b_emb = torch.cat([P[f"conv_{g}.{k}.bias"] for k in range(3) for g in "ab"]).contiguous()
            w_d = torch.zeros((1, 3 * Fp, O), device=dev, dtype=torch.float32)
            for k in range(3):
                w_d[0, k * Fp:k * Fp + Fin] = P[f"conv_d.{k}.weight"].view(O, Fin).t()
            b_d = (P["conv_d.0.bias"] + P["conv_d.1.bias"] + P["conv_d.2.bias"]).contiguous()
            adj_t = (mod.adj_a + adj_b).transpose(1, 2).contiguous()                  # (3, V(w), V(v))
        emb = new(B, V, 6 * ic)
        ops.rows_gemm(_rows4(x), w_emb, _rows4(emb), K=Fp, N=6 * ic, bias=b_emb)
        tp = emb.view(B, V, 6, ic).permute(2, 0, 1, 3).contiguous()                    # (6, B, V, ic): theta_k = tp[2k], phi_k = tp[2k+1]
        th_fm = ops.transpose(tp[0::2].reshape(3 * B, V, ic))                          # (3B, ic, V): theta_k[b]^T as a (K = c, N = v) weight
        st = new(B, 3, V, V)
        # S^T_k[b] = phi_k[b] . theta_k[b]^T: the 3 B problems (b, k) in one launch
        ops.rows_gemm_batched(tp, th_fm, st, batch=B, rows=V, K=ic, N=V, ld_in=ic, ld_out=V, in_bs=V * ic, w_bs=ic * V,
                              out_bs=3 * V * V, in_off=B * V * ic, inner=3, in_bs2=2 * B * V * ic, w_bs2=B * ic * V, out_bs2=V * V)
        c_t, a_t = ops.row_softmax_fwd(st, adj_t, V, 1.0 / ic)
        agg = new(B, V, 3 * Fp)
        # agg_k[b] = A^^T_k[b] . x[b]
        ops.rows_gemm_batched(a_t, x, agg, batch=B, rows=V, K=V, N=Fp, ld_in=V, ld_out=3 * Fp, in_bs=3 * V * V, w_bs=V * Fp,
                              out_bs=V * 3 * Fp, inner=3, in_bs2=V * V, w_bs2=0, out_bs2=Fp)
        y = new(B, V, O)
        part = ops.rows_gemm(_rows4(agg), w_d, _rows4(y), K=3 * Fp, N=O, bias=b_d, stats=train)
        vec_y = (ops.bn_finalize(part, B * V, P["bn.weight"], P["bn.bias"], mod.bn.running_mean, mod.bn.running_var) if train
                 else ops.bn_eval_coeffs(P["bn.weight"], P["bn.bias"], mod.bn.running_mean, mod.bn.running_var))
        d = vec_d = w_down = None
        if mod.has_down:
            with torch.no_grad():
                w_down = torch.zeros((1, Fp, O), device=dev, dtype=torch.float32)
                w_down[0, :Fin] = P["down.0.weight"].view(O, Fin).t()
            d = new(B, V, O)
            part = ops.rows_gemm(_rows4(x), w_down, _rows4(d), K=Fp, N=O, bias=P["down.0.bias"], stats=train)
            bn_d = mod.down[1]
            vec_d = (ops.bn_finalize(part, B * V, P["down.1.weight"], P["down.1.bias"], bn_d.running_mean, bn_d.running_var)
                     if train else ops.bn_eval_coeffs(P["down.1.weight"], P["down.1.bias"], bn_d.running_mean, bn_d.running_var))
            out, mask = ops.bn_act(y, vec_y, d, vec_d, relu=True, sign_mask=True)
        else:
            out, mask = ops.bn_act(y, vec_y, x, None, relu=True, sign_mask=True)
        if train:
            mod.bn.num_batches_tracked += 1
            if mod.has_down:
                mod.down[1].num_batches_tracked += 1
        ctx.mod, ctx.train = mod, train
        ctx.packed = (w_emb, w_d, w_down)
        ctx.save_for_backward(x, tp, c_t, a_t, agg, y, vec_y, d, vec_d, out, mask)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, tp, c_t, a_t, agg, y, vec_y, d, vec_d, out, mask = ctx.saved_tensors
        mod, train = ctx.mod, ctx.train
        w_emb, w_d, w_down = ctx.packed
        B, V, Fp = x.shape
        O, ic, Fin = mod.out_features, mod.inter_c, mod.in_features
        dev = x.device
        new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)      # noqa: E731
        zeros = lambda n: torch.zeros(n, device=dev, dtype=torch.float32)             # noqa: E731
        G = {}
        d_out = d_out.contiguous()
        dx = new(B, V, Fp)
        if mod.has_down:
            dy, dd, sums = ops.bn_act_bwd(d_out, out, y, vec_y, d, vec_d, res_mode=2, train=train, sign_mask=mask)
            G["down.1.weight"], G["down.1.bias"] = sums[2], sums[0].clone()
            ops.rows_gemm(_rows4(dd), w_down[0].t().contiguous().unsqueeze(0), _rows4(dx), K=O, N=Fp)
            G["down.0.weight"] = ops.rows_wgrad(_rows4(x), _rows4(dd), K=Fp, N=O, conv_param=(1, Fin)).view(O, Fin, 1)
            G["down.0.bias"] = zeros(O) if train else ops.col_sum(_rows4(dd), O)
        else:
            dy, _, sums = ops.bn_act_bwd(d_out, out, y, vec_y, x, None, res_mode=1, train=train, db=dx, sign_mask=mask)
        G["bn.weight"], G["bn.bias"] = sums[1], sums[0]
        # conv_d: weights, bias (in front of a BatchNorm: exactly zero in train mode), and dagg_k = dy . Wd_k
        gw = ops.rows_wgrad(_rows4(agg), _rows4(dy), K=3 * Fp, N=O, conv_param=(3, Fin))       # (3, O, Fin, 1, 1)
        dbias = None if train else ops.col_sum(_rows4(dy), O)
        dagg = new(3, B, V, Fp)
        for k in range(3):
            G[f"conv_d.{k}.weight"] = gw[k].view(O, Fin, 1)
            G[f"conv_d.{k}.bias"] = zeros(O) if train else (dbias if k == 0 else dbias.clone())
            wk_t = w_d[0, k * Fp:(k + 1) * Fp].t().contiguous().unsqueeze(0)                     # (1, O, Fp)
            ops.rows_gemm(_rows4(dy), wk_t, _rows4(dagg[k]), K=O, N=Fp)
        # through the aggregation: dx += A^_k . dagg_k,  dA^^T_k = dagg_k . x^T
        a_n = ops.transpose(a_t.view(B * 3, V, V))                                                # (3B, V(v), V(w)): A^_k[b]
        x_fm = ops.transpose(x)                                                                   # (B, Fp, V)
        da_t = new(B, 3, V, V)
        for k in range(3):             # (the three subsets add into the same dx: one after the other)
            ops.rows_gemm_batched(a_n, dagg, dx, batch=B, rows=V, K=V, N=Fp, ld_in=V, ld_out=Fp, in_bs=3 * V * V, w_bs=V * Fp,
                                  out_bs=V * Fp, in_off=k * V * V, w_off=k * B * V * Fp, accumulate=True)
        ops.rows_gemm_batched(dagg, x_fm, da_t, batch=B, rows=V, K=Fp, N=V, ld_in=Fp, ld_out=V, in_bs=V * Fp, w_bs=Fp * V,
                              out_bs=3 * V * V, inner=3, in_bs2=B * V * Fp, w_bs2=0, out_bs2=V * V)
        g_adj_t = new(3, V, V)
        ops.reduce_sum(da_t.view(B, -1), g_adj_t.view(-1))
        G["adj_b"] = g_adj_t.transpose(1, 2).contiguous()
        # softmax, then the embeddings: dphi_k = dS^T_k . theta_k,  dtheta_k = dS_k . phi_k
        ds_t = ops.row_softmax_bwd(da_t, c_t, V, 1.0 / ic)
        ds_n = ops.transpose(ds_t.view(B * 3, V, V))                                              # (3B, V(v), V(w))
        dtp = new(6, B, V, ic)
        # (the 3 B problems (b, k) of each in one launch: theta_k / phi_k are slabs 2k / 2k + 1 of tp and dtp)
        ops.rows_gemm_batched(ds_t, tp, dtp, batch=B, rows=V, K=V, N=ic, ld_in=V, ld_out=ic, in_bs=3 * V * V, w_bs=V * ic,
                              out_bs=V * ic, out_off=B * V * ic, inner=3, in_bs2=V * V, w_bs2=2 * B * V * ic, out_bs2=2 * B * V * ic)
        ops.rows_gemm_batched(ds_n, tp, dtp, batch=B, rows=V, K=V, N=ic, ld_in=V, ld_out=ic, in_bs=3 * V * V, w_bs=V * ic,
                              out_bs=V * ic, w_off=B * V * ic, inner=3, in_bs2=V * V, w_bs2=2 * B * V * ic, out_bs2=2 * B * V * ic)
        demb = dtp.permute(1, 2, 0, 3).reshape(B, V, 6 * ic).contiguous()
        ops.rows_gemm(_rows4(demb), w_emb[0].t().contiguous().unsqueeze(0), _rows4(dx), K=6 * ic, N=Fp, accumulate=True)
        gw = ops.rows_wgrad(_rows4(x), _rows4(demb), K=Fp, N=6 * ic, conv_param=(1, Fin))        # (6ic, Fin, 1, 1)
        gb = ops.col_sum(_rows4(demb), 6 * ic)
        for k in range(3):
            for j, g in enumerate("ab"):
                lo = (2 * k + j) * ic
                G[f"conv_{g}.{k}.weight"] = gw[lo:lo + ic].reshape(ic, Fin, 1)
                G[f"conv_{g}.{k}.bias"] = gb[lo:lo + ic]
        grads = [G[n] for n in mod._param_names()]
        return (dx if ctx.needs_input_grad[0] else None, None, None, G["adj_b"], *grads)


class AGCNGraphConvolution(nn.Module):
    """Same constructor, parameters, initialisation and state-dict keys as the reference class (graph_convolution.py:56-113);
    ``forward`` takes and returns the node-major image (B, V, Fp) like ``STGCNGraphConvolution``."""

    def __init__(self, in_features, out_features, adj, **kwargs):
        super().__init__()
        import numpy as np
        from .agcn import bn_init, conv_branch_init, conv_init
        coff_embedding = kwargs.get("coff_embedding", 4)
        num_subset = kwargs.get("num_subset", 3)
        if num_subset != 3 or out_features % (4 * coff_embedding) or adj.shape[-1] % 4:
            raise ValueError("HIP AGCN graph convolution: 3 subsets, out_features % 16 == 0 and a node count % 4 == 0 "
                             f"(got {num_subset}, {out_features}, {adj.shape[-1]})")
        self.in_features, self.out_features = in_features, out_features
        self.inter_c = out_features // coff_embedding
        self.num_subset = num_subset
        self.adj_b = nn.Parameter(torch.from_numpy(np.asarray(adj).astype(np.float32)))
        nn.init.constant_(self.adj_b, 1e-6)
        self.register_buffer("adj_a", torch.from_numpy(np.asarray(adj).astype(np.float32)))
        self.conv_a, self.conv_b, self.conv_d = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for _ in range(num_subset):
            self.conv_a.append(nn.Conv1d(in_features, self.inter_c, 1))
            self.conv_b.append(nn.Conv1d(in_features, self.inter_c, 1))
            self.conv_d.append(nn.Conv1d(in_features, out_features, 1))
        self.has_down = in_features != out_features
        if self.has_down:
            self.down = nn.Sequential(nn.Conv1d(in_features, out_features, 1), nn.BatchNorm1d(out_features))
        self.bn = nn.BatchNorm1d(out_features)
        for m in self.modules():
            if isinstance(m, nn.Conv1d):
                conv_init(m)
            elif isinstance(m, nn.BatchNorm1d):
                bn_init(m, 1)
        bn_init(self.bn, 1e-6)
        for i in range(num_subset):
            conv_branch_init(self.conv_d[i], num_subset)

    def _param_names(self):
        names = []
        for grp in ("conv_a", "conv_b", "conv_d"):
            for k in range(3):
                names += [f"{grp}.{k}.weight", f"{grp}.{k}.bias"]
        names += ["bn.weight", "bn.bias"]
        if self.has_down:
            names += ["down.0.weight", "down.0.bias", "down.1.weight", "down.1.bias"]
        return names

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        params = dict(self.named_parameters())
        return _AgcnConv1dFunction.apply(x, self, self.training, self.adj_b, *[params[n] for n in self._param_names()])


ops.bind_all_functions(globals())     # every Function's backward runs in its forward's library context (ops.Context)
