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

``AGCNGraphConvolution`` (:56-113, per-sample V x V attention for V in the hundreds) needs batched V-tiled kernels that are
not built yet and raises.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from ... import ops
from ...block import pw_gemm


def _r(c: int, m: int) -> int:
    return (c + m - 1) // m * m


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
            W["w_s3"] = ops.pack_split3(w)
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
                W["wr_s3"] = ops.pack_split3(wr)
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
            wr_t = W["wr"][0].t().contiguous().unsqueeze(0)                      # (1, O, Fp)
            ops.rows_gemm(_rows4(dr), wr_t, _rows4(dx), K=O, N=Fp)
        # main path: d_support = d_main . adj  (feature-major), then the conv's data and weight gradients
        dm_fm = ops.transpose(d_main, Vp)
        ds_fm = torch.empty((B, O, Vp), device=dev, dtype=torch.float32)
        pw_gemm(_rows4(dm_fm), A, "adj", _rows4(ds_fm), K=Vp, N=Vp)
        d_support = ops.transpose_into(ds_fm, V)
        g_w = ops.rows_wgrad(_rows4(x), _rows4(d_support), K=Fp, N=O, conv_param=(1, Fin)).view(O, Fin, 1)
        g_b = ops.col_sum(_rows4(d_support), O)
        if ctx.needs_input_grad[0]:
            w_t = W["w"][0].t().contiguous().unsqueeze(0)                        # (1, O, Fp)
            if dx is None:
                dx = torch.empty_like(x)
                ops.rows_gemm(_rows4(d_support), w_t, _rows4(dx), K=O, N=Fp)
            else:
                ops.rows_gemm(_rows4(d_support), w_t, _rows4(dx), K=O, N=Fp, accumulate=True)
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
                    forms["adjT_s3"] = ops.pack_split3(forms["adjT"])
                    forms["adj_s3"] = ops.pack_split3(forms["adj"])
            self._adj_cache = (key, forms)
        return self._adj_cache[1]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        res = self.residual
        return _GraphConv1dFunction.apply(
            x, self, self.training, self.conv.weight, self.conv.bias,
            res[0].weight if res is not None else None, res[0].bias if res is not None else None,
            res[1].weight if res is not None else None, res[1].bias if res is not None else None)


class AGCNGraphConvolution(nn.Module):
    def __init__(self, in_features, out_features, adj, **kwargs):
        super().__init__()
        raise NotImplementedError("AGCNGraphConvolution (per-sample V x V attention on IMU graphs) is not built yet: "
                                  "use gc_model='stgcn' (SURVEY.md section 8 row f1, DESIGN.md section 0)")
