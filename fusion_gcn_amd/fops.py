"""Op-level autograd wrappers of libfgcn entry points (channels-last activations (B, T, V, C), float32).

The AGCN block is one fused autograd.Function (block.py).  Models whose blocks are compositions of the same kernel families in
other arrangements (MS-G3D: SURVEY.md section 8 row f3) are written like the reference's module code with these differentiable
ops instead; each op is one or a few kernel launches forward and backward, nothing is computed by torch (small parameter
re-layouts aside).  Without libfgcn / off gfx950 every op raises (ops.ensure_device).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .packing import Form, PackPlan, Seg


def _rows4(t: torch.Tensor) -> torch.Tensor:
    """(B, R, C) -> the (B, R, 1, C) view the row kernels index as (sample, frame, joint, channel)."""
    return t.view(t.shape[0], t.shape[1], 1, t.shape[2])


_identity_vecs = {}


def _identity_vec(c: int, device) -> torch.Tensor:
    """(4, C) = {mean 0, rstd 1, scale 1, shift 0}: fgcn_bn_act's coefficients of a tensor without a BatchNorm (read-only, cached)."""
    key = (c, str(device))
    v = _identity_vecs.get(key)
    if v is None:
        v = torch.zeros((4, c), device=device, dtype=torch.float32)
        v[1:3] = 1.0
        _identity_vecs[key] = v
    return v


class deferred_batch_counters:
    """``with deferred_batch_counters() as counters: ...``: the ``num_batches_tracked += 1`` of every BatchNorm that ``bn_act`` runs in
    train mode inside the block is collected and issued as ONE multi-tensor add at exit (117 scalar-add launches per MS-G3D step
    otherwise)."""
    active = None

    def __enter__(self):
        self.buffers = []
        deferred_batch_counters.active = self
        return self

    def __exit__(self, *exc):
        deferred_batch_counters.active = None
        if self.buffers and exc[0] is None:
            torch._foreach_add_(self.buffers, 1)
        return False


class zero_pool:
    """``with zero_pool(model): ...`` around a forward: the exactly-zero gradients of the conv biases in front of train-mode BatchNorms
    (``zero_bias_grad``) are slices of ONE zero-filled allocation per step instead of one fill launch per convolution (the demand
    of a step sizes the next step's pool; every parameter still gets memory of its own)."""
    active = None

    def __init__(self, owner: torch.nn.Module):
        self.owner = owner

    def __enter__(self):
        size = getattr(self.owner, "_zero_pool_size", 0)
        self.buf = torch.zeros(size, device=next(self.owner.parameters()).device, dtype=torch.float32) if size else None
        self.used = self.demand = 0
        zero_pool.active = self
        return self

    def __exit__(self, *exc):
        zero_pool.active = None
        self.owner._zero_pool_size = self.demand
        return False

    @staticmethod
    def take(n: int, device) -> torch.Tensor:
        pool = zero_pool.active
        if pool is None:
            return torch.zeros(n, device=device, dtype=torch.float32)
        pool.demand += n
        if pool.buf is not None and pool.used + n <= pool.buf.numel():
            pool.used += n
            return pool.buf[pool.used - n:pool.used]
        return torch.zeros(n, device=device, dtype=torch.float32)


def _pointwise(tmap) -> bool:
    taps, ta, _tb, tc, td = tmap
    return taps == 1 and ta == 1 and tc == 0 and td == 1


class _ConvRows(torch.autograd.Function):
    """y[(b, to, v), :] = bias + sum_j x[(b, to*ta + j*tb + tc, v), 0:K] . W[j]   (W packed (taps, K, N)); returns (y, BatchNorm
    partial sums of y or an empty tensor).  ``zero_bias_grad``: the bias feeds a train-mode BatchNorm, its gradient is exactly
    zero and is returned as such (the reference's autograd produces rounding noise there)."""

    @staticmethod
    def forward(ctx, x, w, bias, tmap, T_out: int, stats: bool, zero_bias_grad: bool):
        B, T, V, ld = x.shape
        taps, K, N = w.shape
        out = torch.empty((B, T_out, V, N), device=x.device, dtype=torch.float32)
        if _pointwise(tmap):                        # rows are rows: fold the node axis into the frames (node counts beyond the
            x = x.view(B, T * V, 1, ld)             # 32-joint limit of the temporal kernels, e.g. MS-G3D's 135-node windows)
            part = ops.rows_gemm(x, w.contiguous(), out.view(B, T * V, 1, N), K=K, N=N, bias=bias, stats=stats)
        else:
            part = ops.rows_gemm(x, w.contiguous(), out, K=K, N=N, tmap=tmap, bias=bias, stats=stats)
        ctx.save_for_backward(x, w)
        ctx.tmap, ctx.has_bias, ctx.zero_bias_grad = tmap, bias is not None, zero_bias_grad
        if part is None:
            part = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)     # (no zero tensor for `part` in the backward)
        return out, part

    @staticmethod
    def backward(ctx, d_out, _d_part):
        if d_out is None:                        # (unused output)
            return (None,) * 7
        x, w = ctx.saved_tensors
        taps, ta, tb, tc, td = ctx.tmap
        _, K, N = w.shape
        d_out = d_out.contiguous()
        shape_out = None
        if _pointwise(ctx.tmap):                    # x was saved in its folded (B, T*V, 1, ld) form
            shape_out = (d_out.shape[0], d_out.shape[1], d_out.shape[2], x.shape[3])
            d_out = d_out.view(x.shape[0], x.shape[1], 1, N)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if x.shape[3] != K:
                dx.zero_()                                           # channels beyond the K window receive nothing
            ops.rows_gemm(d_out, w.transpose(1, 2).contiguous(), dx, K=N, N=K, tmap=(taps, td, -tb, -tc, ta))
        if ctx.needs_input_grad[1]:
            dw = ops.rows_wgrad(x, d_out, K=K, N=N, tmap=ctx.tmap)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.zeros(N, device=x.device, dtype=torch.float32) if ctx.zero_bias_grad else ops.col_sum(d_out, N)
        if dx is not None and shape_out is not None:
            dx = dx.view(shape_out)
        return dx, dw, db, None, None, None, None


def conv_rows(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], tmap=ops.TMAP_POINTWISE, T_out: Optional[int] = None,
              stats: bool = False, zero_bias_grad: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    return _ConvRows.apply(x, w, bias, tuple(tmap), x.shape[1] if T_out is None else T_out, stats, zero_bias_grad)


class _BnAct(torch.autograd.Function):
    """act(BatchNorm(a) [+ res]) with batch statistics from the producer's partial sums (train) or the running ones (eval)."""

    @staticmethod
    def forward(ctx, a, part, gamma, beta, running_mean, running_var, train: bool, res, relu: bool, eps: float, momentum: float):
        C = a.shape[-1]
        count = a.numel() // C
        if train:
            vec = ops.bn_finalize(part, count, gamma, beta, running_mean, running_var, momentum=momentum, eps=eps)
        else:
            vec = ops.bn_eval_coeffs(gamma, beta, running_mean, running_var, eps=eps)
        if relu:
            out, mask = ops.bn_act(a, vec, res, None, relu=True, sign_mask=True)
        else:
            out, mask = ops.bn_act(a, vec, res, None, relu=False), None
        ctx.train, ctx.relu, ctx.has_res = train, relu, res is not None
        ctx.save_for_backward(a, vec, mask, out if (relu and mask is None) else None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        a, vec, mask, out = ctx.saved_tensors
        d_out = d_out.contiguous()
        da, dres, sums = ops.bn_act_bwd(d_out, out, a, vec, a if ctx.has_res else None, None, relu=ctx.relu, train=ctx.train,
                                        res_mode=1 if ctx.has_res else 0, sign_mask=mask)
        return da, None, sums[1], sums[0], None, None, None, (dres if ctx.has_res else None), None, None, None


def bn_act(a: torch.Tensor, part: torch.Tensor, bn: torch.nn.Module, res: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """``bn``: an nn.BatchNorm2d used as the parameter / buffer container (its own forward is never called)."""
    train = bn.training
    if train and isinstance(bn.num_batches_tracked, torch.Tensor):
        if deferred_batch_counters.active is not None:
            deferred_batch_counters.active.buffers.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
    return _BnAct.apply(a, part, bn.weight, bn.bias, bn.running_mean, bn.running_var, train, res, relu, bn.eps,
                        0.1 if bn.momentum is None else bn.momentum)


class _BnCat(torch.autograd.Function):
    """cat_i BatchNorm_i(a_i) on the channel axis as ONE node: every branch's BatchNorm writes its channel window of the result
    (ops.bn_apply_window) and, in the backward, reads its window of the result's gradient in place (ops.bn_bwd_window) -- no
    torch.cat, no contiguous copies of the cat's backward slices.  args: n tensors a_i, n partials, n gammas, n betas, n running
    means, n running variances (all branches: C channels, no residual, no activation)."""

    @staticmethod
    def forward(ctx, n: int, train: bool, eps: float, momentum: float, *args):
        a, part, gamma, beta, rmean, rvar = (args[i * n:(i + 1) * n] for i in range(6))
        C = a[0].shape[-1]
        count = a[0].numel() // C
        out = torch.empty((*a[0].shape[:-1], n * C), device=a[0].device, dtype=torch.float32)
        vecs = []
        for i in range(n):
            if train:
                vec = ops.bn_finalize(part[i], count, gamma[i], beta[i], rmean[i], rvar[i], momentum=momentum, eps=eps)
            else:
                vec = ops.bn_eval_coeffs(gamma[i], beta[i], rmean[i], rvar[i], eps=eps)
            ops.bn_apply_window(a[i], vec, out, i * C)
            vecs.append(vec)
        ctx.n, ctx.train, ctx.C = n, train, C
        ctx.save_for_backward(*a, *vecs)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, d_out):
        n, C = ctx.n, ctx.C
        saved = ctx.saved_tensors
        a, vecs = saved[:n], saved[n:]
        if d_out is None:
            return (None,) * (4 + 6 * n)
        d_out = d_out.contiguous()
        das, dgs, dbs = [], [], []
        for i in range(n):
            da, sums = ops.bn_bwd_window(d_out, i * C, a[i], vecs[i], train=ctx.train)
            das.append(da), dgs.append(sums[1]), dbs.append(sums[0])
        return (None, None, None, None, *das, *([None] * n), *dgs, *dbs, *([None] * (2 * n)))


def bn_cat(branches) -> torch.Tensor:
    """``branches``: [(a_i, partials_i, bn_i)] with equal channel counts, BatchNorm parameters and mode -> cat_i bn_i(a_i) on the
    channel axis (the six branches of MS-G3D's multi-scale temporal convolution, ms_tcn.py:88-109)."""
    bns = [b for _, _, b in branches]
    train = bns[0].training
    for bn in bns:
        if train and isinstance(bn.num_batches_tracked, torch.Tensor):
            if deferred_batch_counters.active is not None:
                deferred_batch_counters.active.buffers.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked += 1
    n = len(branches)
    mom = 0.1 if bns[0].momentum is None else bns[0].momentum
    return _BnCat.apply(n, train, bns[0].eps, mom, *[a.contiguous() for a, _, _ in branches], *[p for _, p, _ in branches],
                        *[b.weight for b in bns], *[b.bias for b in bns], *[b.running_mean for b in bns], *[b.running_var for b in bns])


class _AddRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, relu: bool):
        vec = _identity_vec(a.shape[-1], a.device)
        if relu:
            out, mask = ops.bn_act(a, vec, b, None, relu=True, sign_mask=True)
        else:
            out, mask = ops.bn_act(a, vec, b, None, relu=False), None
        ctx.relu = relu
        ctx.save_for_backward(a, vec, mask, out if (relu and mask is None) else None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        a, vec, mask, out = ctx.saved_tensors
        d_out = d_out.contiguous()
        if not ctx.relu:
            return d_out, d_out, None
        da, db, _ = ops.bn_act_bwd(d_out, out, a, vec, a, None, relu=True, train=False, res_mode=1, sign_mask=mask, need_sums=False)
        return da, db, None


def add_act(a: torch.Tensor, b: torch.Tensor, relu: bool = True) -> torch.Tensor:
    """act(a + b) in one pass (fgcn_bn_act with identity coefficients); the backward gates both gradients from the sign image."""
    return _AddRelu.apply(a.contiguous(), b.contiguous(), relu)


class _NodeMix(torch.autograd.Function):
    """Static multi-scale aggregation over the node axis: out[(b, t, v), s*C + c] = sum_u A[s*V + v, u] x[(b, t, u), c].

    ``a_fm`` (Vp, Np) is the matrix in the layout the kernel contracts with: a_fm[u, v*S + s] = A[s*V + v, u], zero-padded to
    Vp = V rounded up to 64 rows and Np = V*S rounded up to 4 columns.  The node axis goes through the feature-major image
    (fgcn_transpose), the contraction is the row GEMM over the B*T*C feature rows with a_fm as the SHARED weight -- any number
    of nodes (the spatial-temporal windows of MS-G3D have up to 135), unlike the <= 32-joint register kernels of the AGCN block."""

    @staticmethod
    def forward(ctx, x, a_fm, S: int):
        B, T, V, C = x.shape
        Vp, Np = a_fm.shape
        x_fm = ops.transpose(x.view(B * T, V, C), Vp)                              # (BT, C, Vp), zero padding columns
        out_fm = torch.empty((B * T, C, Np), device=x.device, dtype=torch.float32)
        ops.rows_gemm(_rows4(x_fm), a_fm.contiguous().unsqueeze(0), _rows4(out_fm), K=Vp, N=Np)
        out = ops.transpose_into(out_fm, V * S)                                     # (BT, V*S, C)
        ctx.save_for_backward(x_fm, a_fm)
        ctx.dims = (B, T, V, C, S)
        return out.view(B, T, V, S * C)

    @staticmethod
    def backward(ctx, d_out):
        x_fm, a_fm = ctx.saved_tensors
        B, T, V, C, S = ctx.dims
        Vp, Np = a_fm.shape
        d_fm = ops.transpose(d_out.contiguous().view(B * T, V * S, C), Np)          # (BT, C, Np)
        dx = da = None
        if ctx.needs_input_grad[0]:
            dx_fm = torch.empty((B * T, C, Vp), device=d_out.device, dtype=torch.float32)
            ops.rows_gemm(_rows4(d_fm), a_fm.t().contiguous().unsqueeze(0), _rows4(dx_fm), K=Np, N=Vp)
            dx = ops.transpose_into(dx_fm, V).view(B, T, V, C)
        if ctx.needs_input_grad[1]:
            da = ops.rows_wgrad(_rows4(x_fm), _rows4(d_fm), K=Vp, N=Np, wide=False)[0]
        return dx, da, None


def node_mix(x: torch.Tensor, a_fm: torch.Tensor, num_scales: int) -> torch.Tensor:
    return _NodeMix.apply(x.contiguous(), a_fm, num_scales)


def node_mix_matrix(a: torch.Tensor, num_scales: int) -> torch.Tensor:
    """(S*V, V) stacked adjacency (differentiable: the learnable residual is part of it) -> the (Vp, Np) form ``node_mix`` takes."""
    SV, V = a.shape
    S = num_scales
    assert SV == S * V
    fm = a.view(S, V, V).permute(2, 1, 0).reshape(V, V * S)                         # [u, v*S + s] = A[s*V + v, u]
    Vp, Np = (V + 63) // 64 * 64, (V * S + 3) // 4 * 4
    return torch.nn.functional.pad(fm, (0, Np - V * S, 0, Vp - V))


class _WindowBranches(torch.autograd.Function):
    """The temporal branches of MS-G3D's MultiScale_TemporalConv over ONE shared head tensor h (B, T, V, (n+1)*bc): branch i < n is a
    (k x 1) convolution with dilation d_i and the block's stride on the channel window [i*bc, (i+1)*bc) of h, the last window goes
    through the (3 x 1) max pooling.  The kernels read their windows in place (row stride of h) and the backward writes every
    branch's input gradient into its window of ONE gradient tensor -- no per-branch slice copies, zero-filled full-size gradients
    or gradient adds (19 launches less per block than slicing h with torch).
    Inputs: h, then n packed weights (k, bc, bc), then n biases.  Outputs: n conv outputs, the pooled window, n BatchNorm partials."""

    @staticmethod
    def forward(ctx, h, n: int, bc: int, tmaps, stride: int, T_out: int, stats: bool, zero_bias_grad: bool, *wb):
        ws, bs = wb[:n], wb[n:]
        B, T, V, ld = h.shape
        outs, parts = [], []
        for i in range(n):
            y = torch.empty((B, T_out, V, bc), device=h.device, dtype=torch.float32)
            part = ops.rows_gemm(h, ws[i].contiguous(), y, K=bc, N=bc, tmap=tmaps[i], bias=bs[i], stats=stats, in_coff=i * bc)
            outs.append(y)
            parts.append(part if part is not None else torch.empty(0, device=h.device))
        pooled, idx = ops.tmaxpool3_fwd(h, stride, coff=n * bc, C=bc)
        ctx.save_for_backward(h, idx, *ws)
        ctx.cfg = (n, bc, tmaps, stride, zero_bias_grad)
        ctx.mark_non_differentiable(*parts)
        return (*outs, pooled, *parts)

    @staticmethod
    def backward(ctx, *grads):
        h, idx, *ws = ctx.saved_tensors
        n, bc, tmaps, stride, zero_bias_grad = ctx.cfg
        B, T, V, ld = h.shape
        d_outs, d_pool = grads[:n], grads[n]
        dh = torch.empty_like(h)
        gws, gbs = [], []
        zeros = torch.zeros(n * bc, device=h.device, dtype=torch.float32) if zero_bias_grad else None
        for i in range(n):
            d = d_outs[i].contiguous()
            taps, ta, tb, tc, td = tmaps[i]
            ops.rows_gemm(d, ws[i].transpose(1, 2).contiguous(), dh, K=bc, N=bc, tmap=(taps, td, -tb, -tc, ta), out_coff=i * bc)
            gws.append(ops.rows_wgrad(h, d, K=bc, N=bc, tmap=tmaps[i], a_coff=i * bc))
            gbs.append(zeros[i * bc:(i + 1) * bc] if zero_bias_grad else ops.col_sum(d, bc))
        ops.tmaxpool3_bwd(d_pool.contiguous(), idx, T, stride, din=dh, coff=n * bc)
        return (dh, None, None, None, None, None, None, None, *gws, *gbs)


def window_branches(h, weights, biases, tmaps, bc: int, stride: int, T_out: int, stats: bool, zero_bias_grad: bool):
    n = len(weights)
    out = _WindowBranches.apply(h.contiguous(), n, bc, tuple(tuple(t) for t in tmaps), stride, T_out, stats, zero_bias_grad, *weights, *biases)
    return out[:n], out[n], out[n + 1:]


class _MaxPool3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride: int):
        out, idx = ops.tmaxpool3_fwd(x, stride)
        ctx.save_for_backward(idx)
        ctx.T, ctx.stride = x.shape[1], stride
        return out

    @staticmethod
    def backward(ctx, d_out):
        (idx,) = ctx.saved_tensors
        return ops.tmaxpool3_bwd(d_out.contiguous(), idx, ctx.T, ctx.stride), None


def maxpool3(x: torch.Tensor, stride: int) -> torch.Tensor:
    return _MaxPool3.apply(x.contiguous(), stride)


class _Unfold(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, window: int, stride: int, dilation: int):
        ctx.args = (x.shape[1], x.shape[2], window, stride, dilation)
        return ops.unfold_windows(x, window, stride, dilation)

    @staticmethod
    def backward(ctx, d_out):
        T, V, window, stride, dilation = ctx.args
        return ops.unfold_windows_bwd(d_out.contiguous(), T, V, window, stride, dilation), None, None, None


def unfold_windows(x: torch.Tensor, window: int, stride: int, dilation: int = 1) -> torch.Tensor:
    return _Unfold.apply(x.contiguous(), window, stride, dilation)


def col_stats(x: torch.Tensor) -> torch.Tensor:
    """BatchNorm partial sums of a tensor that no GEMM epilogue produced (the max-pooled branch); not differentiated (the
    BatchNorm backward works from the tensor itself)."""
    with torch.no_grad():
        return ops.col_moments(x.detach().contiguous())


# ---- parameters in the layouts the kernels stream, re-packed for a whole model in one launch -----------------------------------
# The ops above take weights already packed (taps, K, N); building that from an nn.Conv2d parameter with torch ops (reshape /
# permute / contiguous / pad, and their backward) is ~10 tiny launches per convolution and step -- ~800 of MS-G3D's 1650.  Below the
# packed matrices (and their transposes for the data gradients) are packing.Forms over the parameters, all of a model refreshed by
# ONE fgcn_pack_run launch per step (refresh_forms), and the weight gradients come out of the wgrad kernels' reduction directly in
# the parameter's own layout (ops.rows_wgrad(conv_param=...)): no torch op touches a weight.

class ParamForms:
    """The packed forms of one module's parameters, created (and packed once on their own) on first use.  Every hand-out checks
    the form against its sources: parameters updated in place since the last pack (an optimizer step, load_state_dict) are
    re-packed, parameters that moved (module.to()) get a new form -- a module used on its own behaves like one under a Model
    whose forward runs the batched ``refresh_forms`` (which leaves every form it packs marked fresh)."""

    def __init__(self):
        self.forms: Dict[str, Form] = {}
        self.generation = 0

    def _current(self, key: str) -> Optional[Form]:
        f = self.forms.get(key)
        if f is None:
            return None
        now = f.stamp()
        if f.packed == now:
            return f
        if f.packed is None or any(a[0] != b[0] for a, b in zip(f.packed, now)):
            del self.forms[key]                      # a source moved: the form's table holds dead addresses
            self.generation += 1
            return None
        solo = getattr(f, "_solo_plan", None)
        if solo is None:
            solo = f._solo_plan = PackPlan([f], stamp=True)
        solo.run()
        return f

    def get(self, name: str, make: Callable[[], Form], device) -> torch.Tensor:
        f = self._current(name)
        if f is None:
            f = make()
            f.alloc(device)
            f._solo_plan = PackPlan([f], stamp=True)
            f._solo_plan.run()
            self.forms[name] = f
            self.generation += 1
        return f.dst

    def get_pair(self, name: str, suffixes: Tuple[str, str], make: Callable[[], Tuple[Form, Form]], device):
        """Two forms built together (a matrix and its transpose) under ``name + suffix``."""
        a, b = self._current(name + suffixes[0]), self._current(name + suffixes[1])
        if a is None or b is None:
            a, b = make()
            for key, f in ((name + suffixes[0], a), (name + suffixes[1], b)):
                f.alloc(device)
                self.forms[key] = f
            PackPlan([a, b], stamp=True).run()
            self.generation += 1
        return a.dst, b.dst

    def clear(self) -> None:
        self.forms.clear()
        self.generation += 1


def refresh_forms(model: torch.nn.Module) -> None:
    """Call at the top of a model's forward: re-pack every existing form of every sub-module when a parameter changed since the last
    pack (version counters), in one launch; forms whose parameters moved (model.to(), an optimizer's flat home) are dropped and
    rebuilt by their modules on use."""
    sets = [m._forms for m in model.modules() if isinstance(getattr(m, "_forms", None), ParamForms)]
    gen = sum(s.generation for s in sets)
    if not any(s.forms for s in sets):
        return
    params = list(model.parameters())
    homes = tuple(p.data_ptr() for p in params)
    state = getattr(model, "_forms_state", None)
    if state is not None and state["homes"] != homes:
        for s_ in sets:
            s_.clear()
        model._forms_state = None
        return
    if state is None or state["gen"] != gen:
        state = {"gen": gen, "homes": homes, "plan": PackPlan([f for s_ in sets for f in s_.forms.values()], stamp=True), "versions": None}
        model._forms_state = state
    versions = tuple(p._version for p in params)
    if state["versions"] != versions:
        state["plan"].run()
        state["versions"] = versions


def recording_pins(model: torch.nn.Module) -> list:
    """What a HIP-graph recording of ``model`` reads through raw pointers: every module's forms and the model's re-pack plan
    (GraphStep keeps them alive with the recording)."""
    pins = [f for m in model.modules() if isinstance(getattr(m, "_forms", None), ParamForms) for f in m._forms.forms.values()]
    state = getattr(model, "_forms_state", None)
    if state is not None:
        pins.append(state["plan"])
    return pins


def mark_forms_stale(model: torch.nn.Module) -> None:
    state = getattr(model, "_forms_state", None)
    if state is not None:
        state["versions"] = None


def conv_weight_forms(weights: Sequence[torch.Tensor], k_pad: int = 0) -> Tuple[Form, Form]:
    """Forms of the (taps, K + k_pad, sum N_i) matrix of Conv weights (O_i, I, taps[, 1]) concatenated along the output channels, and
    of its per-tap transpose (taps, sum N_i, K + k_pad) for the data gradient; the k_pad input channels are zero rows."""
    o_all = sum(w.shape[0] for w in weights)
    inner = weights[0].shape[1]
    taps = weights[0].numel() // (weights[0].shape[0] * inner)
    fwd, bwd, n0 = [], [], 0
    for w in weights:
        o = w.shape[0]
        fwd.append(Seg(w, st_k=taps, st_n=inner * taps, klen=inner, nlen=o, n0=n0, st_tap=1, tlen=taps))
        bwd.append(Seg(w, st_k=inner * taps, st_n=taps, klen=o, nlen=inner, k0=n0, st_tap=1, tlen=taps))
        n0 += o
    return Form("plain", taps, inner + k_pad, o_all, fwd), Form("plain", taps, o_all, inner + k_pad, bwd)


def scale_major_forms(weight: torch.Tensor, num_scales: int, c_pad: int) -> Tuple[Form, Form]:
    """MLP weight (O, S*C[, 1, 1]) whose input channel is s*C + c, for an aggregate whose per-scale channel groups are C + c_pad wide:
    the (1, S*(C + c_pad), O) matrix stated as (S, C + c_pad, O) -- one "tap" per scale -- and its transpose (O, S, C + c_pad)."""
    o = weight.shape[0]
    c = weight.numel() // (o * num_scales)
    fwd = Form("plain", num_scales, c + c_pad, o, [Seg(weight, st_k=1, st_n=num_scales * c, klen=c, nlen=o, st_tap=c, tlen=num_scales)],
               shape=(1, num_scales * (c + c_pad), o))
    bwd = Form("plain", o, num_scales, c + c_pad, [Seg(weight, st_k=c, st_n=1, klen=num_scales, nlen=c, st_tap=num_scales * c, tlen=o)],
               shape=(1, o, num_scales * (c + c_pad)))
    return fwd, bwd


def bias_form(biases: Sequence[Optional[torch.Tensor]]) -> Optional[Form]:
    """The concatenation of the convolutions' biases as one vector (None when no convolution has one)."""
    if all(b is None for b in biases):
        return None
    segs, n0 = [], 0
    for b in biases:
        segs.append(Seg(b, st_k=0, st_n=1, klen=1, nlen=b.numel(), n0=n0))
        n0 += b.numel()
    return Form("plain", 1, 1, n0, segs, shape=(n0,))


def node_mix_forms(a_const: torch.Tensor, a_res: torch.Tensor, num_scales: int) -> Tuple[Form, Form]:
    """node_mix's matrix a_fm[u, v*S + s] = (A + A_res)[s*V + v, u] (Vp x V*S, see node_mix_matrix) and its transpose, as sums of the
    constant stack and the learnable residual: (taps, K, N) = (u, v, s) resp. (v, s, u) index the same memory."""
    SV, V = a_const.shape
    S = num_scales
    if SV != S * V or (V * S) % 4:
        raise ValueError("node_mix_forms: stacked (S*V, V) matrix with V*S a multiple of 4 expected")
    Vp = (V + 63) // 64 * 64
    fwd = Form("plain", Vp, V, S, [Seg(t, st_k=V, st_n=V * V, klen=V, nlen=S, st_tap=1, tlen=V) for t in (a_const, a_res)],
               shape=(Vp, V * S))
    bwd = Form("plain", V, S, Vp, [Seg(t, st_k=V * V, st_n=1, klen=S, nlen=V, st_tap=V, tlen=V) for t in (a_const, a_res)],
               shape=(V * S, Vp))
    return fwd, bwd


class _ConvParams(torch.autograd.Function):
    """_ConvRows on parameters: ``w`` / ``wt`` / ``bias`` are packed forms of ``weights`` / ``biases`` (refresh_forms keeps them current);
    gradients are returned per parameter, in the parameter's layout, straight from the weight-gradient reduction.
    ``groups`` > 1: the matrix is `groups` stacked (K/groups x N) blocks whose parameter is laid out (O, groups * k_true) (the
    scale-major MLPs behind node_mix)."""

    @staticmethod
    def forward(ctx, x, w, wt, bias, cfg, *params):
        tmap, T_out, stats, zero_bias_grad, in_coff, k_true, groups, n_w = cfg
        B, T, V, ld = x.shape
        taps, K, N = w.shape
        out = torch.empty((B, T_out, V, N), device=x.device, dtype=torch.float32)
        fold = _pointwise(tmap) and in_coff == 0
        xin = x.view(B, T * V, 1, ld) if fold else x
        part = ops.rows_gemm(xin, w, out.view(B, T * V, 1, N) if fold else out, K=K, N=N, tmap=tmap, bias=bias, stats=stats,
                             in_coff=in_coff)
        ctx.save_for_backward(xin, wt)
        ctx.cfg, ctx.fold, ctx.x_shape = cfg, fold, x.shape
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.zero_bias = zero_pool.take(N, x.device) if (zero_bias_grad and len(params) > n_w) else None
        if part is None:
            part = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)     # (no zero tensor for `part` in the backward: one fill launch per convolution and step)
        return out, part

    @staticmethod
    def backward(ctx, d_out, _d_part):
        xin, wt = ctx.saved_tensors
        tmap, T_out, stats, zero_bias_grad, in_coff, k_true, groups, n_w = ctx.cfg
        taps, ta, tb, tc, td = tmap
        _, N, K = wt.shape
        B, T, V, ld = ctx.x_shape
        if d_out is None:                        # (unused output; gradients are not materialised, see forward)
            d_out = torch.zeros((B, T_out, V, N), device=xin.device, dtype=torch.float32)
        d_out = d_out.contiguous()
        if ctx.fold:
            d_out = d_out.view(B, T * V, 1, N)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(xin)
            if in_coff != 0 or ld != K:
                dx.zero_()                                              # channels outside the K window receive nothing
            ops.rows_gemm(d_out, wt, dx, K=N, N=K, tmap=(taps, td, -tb, -tc, ta), out_coff=in_coff)
            dx = dx.view(ctx.x_shape)
        shapes = ctx.shapes
        grads: List[Optional[torch.Tensor]] = [None] * len(shapes)
        if any(ctx.needs_input_grad[5:5 + n_w]):
            gw = ops.rows_wgrad(xin, d_out, K=K, N=N, tmap=tmap, a_coff=in_coff, conv_param=(groups, k_true))
            if groups > 1:                                              # (groups, N, k_true, 1, 1) -> the parameter's (N, groups * k_true)
                gw = gw.view(groups, N, k_true).permute(1, 0, 2).reshape(N, groups * k_true)
            n0 = 0
            for i in range(n_w):
                o = shapes[i][0]
                grads[i] = gw[n0:n0 + o].view(shapes[i])
                n0 += o
        if len(shapes) > n_w and any(ctx.needs_input_grad[5 + n_w:]):
            gb = ctx.zero_bias if zero_bias_grad else ops.col_sum(d_out, N)
            n0 = 0
            for i in range(n_w, len(shapes)):
                o = shapes[i][0]
                grads[i] = gb[n0:n0 + o]
                n0 += o
        return (dx, None, None, None, None, *grads)


def conv_params(x: torch.Tensor, forms: ParamForms, name: str, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]],
                *, tmap=ops.TMAP_POINTWISE, T_out: Optional[int] = None, stats: bool = False, zero_bias_grad: bool = False,
                in_coff: int = 0, scales: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """The convolution(s) ``weights`` (concatenated along their output channels; ``scales`` > 1: ONE scale-major MLP weight) applied to
    the channel window of x that starts at ``in_coff`` -> (y, BatchNorm partial sums or an empty tensor).  The packed forms live in
    ``forms`` under ``name``."""
    weights, biases = list(weights), [b for b in biases if b is not None]
    x = x.contiguous()
    if scales > 1 and x.shape[-1] == weights[0].shape[1]:
        scales = 1                                                      # unpadded scale groups: an ordinary (O, S*C) matrix
    inner = weights[0].shape[1] // scales
    k_pad = x.shape[-1] // scales - inner if scales > 1 else 0
    if scales == 1 and in_coff == 0 and x.shape[-1] != inner:
        k_pad = x.shape[-1] - inner                                     # zero pad channels of the input (3 -> 4)
    w, wt = forms.get_pair(name, (".w", ".wt"), lambda: (scale_major_forms(weights[0], scales, k_pad) if scales > 1
                                                          else conv_weight_forms(weights, k_pad)), x.device)
    bias = forms.get(name + ".b", lambda: bias_form(biases), x.device) if biases else None
    cfg = (tuple(tmap), x.shape[1] if T_out is None else T_out, stats, zero_bias_grad, in_coff, inner, scales, len(weights))
    return _ConvParams.apply(x, w, wt, bias, cfg, *weights, *biases)


class _NodeMixParams(torch.autograd.Function):
    """_NodeMix with the packed matrix and its transpose as forms of (A + A_res); returns the gradient of A_res in its own layout."""

    @staticmethod
    def forward(ctx, x, a_fm, a_fm_t, a_res, S: int):
        B, T, V, C = x.shape
        Vp, Np = a_fm.shape
        x_fm = ops.transpose(x.view(B * T, V, C), Vp)
        out_fm = torch.empty((B * T, C, Np), device=x.device, dtype=torch.float32)
        ops.rows_gemm(_rows4(x_fm), a_fm.unsqueeze(0), _rows4(out_fm), K=Vp, N=Np)
        out = ops.transpose_into(out_fm, V * S)
        ctx.save_for_backward(x_fm, a_fm_t)
        ctx.dims = (B, T, V, C, S)
        return out.view(B, T, V, S * C)

    @staticmethod
    def backward(ctx, d_out):
        x_fm, a_fm_t = ctx.saved_tensors
        B, T, V, C, S = ctx.dims
        Np, Vp = a_fm_t.shape
        d_fm = ops.transpose(d_out.contiguous().view(B * T, V * S, C), Np)
        dx = da = None
        if ctx.needs_input_grad[0]:
            dx_fm = torch.empty((B * T, C, Vp), device=d_out.device, dtype=torch.float32)
            ops.rows_gemm(_rows4(d_fm), a_fm_t.unsqueeze(0), _rows4(dx_fm), K=Np, N=Vp)
            dx = ops.transpose_into(dx_fm, V).view(B, T, V, C)
        if ctx.needs_input_grad[3]:
            da_fm = ops.rows_wgrad(_rows4(x_fm), _rows4(d_fm), K=Vp, N=Np, wide=False)[0]        # (Vp, V*S): [u, v*S + s]
            da = da_fm[:V].view(V, V, S).permute(2, 1, 0).reshape(S * V, V)
        return dx, None, None, da, None


def node_mix_params(x: torch.Tensor, forms: ParamForms, name: str, a_const: torch.Tensor, a_res: torch.Tensor, num_scales: int):
    """node_mix with the stacked matrix (a_const + a_res); the packed matrix and its transpose are forms when V * S needs no column
    padding (UTD-MHAD's 20 joints: every case), built with torch ops otherwise (NTU's 25 joints x 13 scales = 325 columns)."""
    if (a_const.shape[1] * num_scales) % 4:
        return node_mix(x, node_mix_matrix(a_const + a_res, num_scales), num_scales)
    a_fm, a_fm_t = forms.get_pair(name, (".a", ".at"), lambda: node_mix_forms(a_const, a_res, num_scales), x.device)
    return _NodeMixParams.apply(x.contiguous(), a_fm, a_fm_t, a_res, num_scales)


class _WindowBranchesParams(torch.autograd.Function):
    """_WindowBranches on parameters: inputs h, the n packed weights, their n transposes, the n packed biases, then the n weight and n
    bias parameters (for the gradients, which come back in the parameters' layouts)."""

    @staticmethod
    def forward(ctx, h, cfg, *rest):
        n, bc, tmaps, stride, T_out, stats, zero_bias_grad = cfg
        ws, wts, bs = rest[:n], rest[n:2 * n], rest[2 * n:3 * n]
        B, T, V, ld = h.shape
        outs, parts = [], []
        for i in range(n):
            y = torch.empty((B, T_out, V, bc), device=h.device, dtype=torch.float32)
            part = ops.rows_gemm(h, ws[i], y, K=bc, N=bc, tmap=tmaps[i], bias=bs[i], stats=stats, in_coff=i * bc)
            outs.append(y)
            parts.append(part if part is not None else torch.empty(0, device=h.device))
        pooled, idx = ops.tmaxpool3_fwd(h, stride, coff=n * bc, C=bc)
        ctx.save_for_backward(h, idx, *wts)
        ctx.cfg = cfg
        ctx.shapes = [tuple(p.shape) for p in rest[3 * n:]]
        ctx.zero_bias = zero_pool.take(n * bc, h.device) if zero_bias_grad else None
        ctx.mark_non_differentiable(*parts)
        ctx.set_materialize_grads(False)     # (no zero tensors for the partial sums in the backward)
        return (*outs, pooled, *parts)

    @staticmethod
    def backward(ctx, *grads):
        h, idx, *wts = ctx.saved_tensors
        n, bc, tmaps, stride, T_out, stats, zero_bias_grad = ctx.cfg
        B, T, V, ld = h.shape
        d_outs, d_pool = list(grads[:n]), grads[n]
        dh = torch.empty_like(h)
        gws, gbs = [], []
        zeros = ctx.zero_bias
        for i in range(n):
            if d_outs[i] is None:                # (unused output; gradients are not materialised, see forward)
                d_outs[i] = torch.zeros((B, T_out, V, bc), device=h.device, dtype=torch.float32)
            d = d_outs[i].contiguous()
            taps, ta, tb, tc, td = tmaps[i]
            ops.rows_gemm(d, wts[i], dh, K=bc, N=bc, tmap=(taps, td, -tb, -tc, ta), out_coff=i * bc)
            gws.append(ops.rows_wgrad(h, d, K=bc, N=bc, tmap=tmaps[i], a_coff=i * bc, conv_param=(1, bc)).view(ctx.shapes[i]))
            gbs.append(zeros[i * bc:(i + 1) * bc] if zero_bias_grad else ops.col_sum(d, bc))
        if d_pool is None:
            d_pool = torch.zeros((B, T_out, V, bc), device=h.device, dtype=torch.float32)
        ops.tmaxpool3_bwd(d_pool.contiguous(), idx, T, stride, din=dh, coff=n * bc)
        return (dh, None, *([None] * (3 * n)), *gws, *gbs)


def window_branches_params(h, forms: ParamForms, name: str, convs, tmaps, bc: int, stride: int, T_out: int, stats: bool,
                           zero_bias_grad: bool):
    """``convs``: the n nn.Conv2d of the dilated branches (weights (bc, bc, k, 1))."""
    n = len(convs)
    ws, wts, bs = [], [], []
    for i, conv in enumerate(convs):
        w, wt = forms.get_pair(f"{name}.{i}", (".w", ".wt"), lambda conv=conv: conv_weight_forms([conv.weight]), h.device)
        ws.append(w)
        wts.append(wt)
        bs.append(forms.get(f"{name}.{i}.b", lambda conv=conv: bias_form([conv.bias]), h.device))
    cfg = (n, bc, tuple(tuple(t) for t in tmaps), stride, T_out, stats, zero_bias_grad)
    out = _WindowBranchesParams.apply(h.contiguous(), cfg, *ws, *wts, *bs, *[c.weight for c in convs], *[c.bias for c in convs])
    return out[:n], out[n], out[n + 1:]


class _JoinedVector(torch.autograd.Function):
    """The concatenation of 1-D parameters as ONE packed form (bias_form; refreshed with the model's other forms): forward hands
    out the packed buffer, backward the slices of its gradient -- no cat / split launches."""

    @staticmethod
    def forward(ctx, packed, *params):
        ctx.sizes = [p.numel() for p in params]
        return packed.view_as(packed)

    @staticmethod
    def backward(ctx, g):
        out, lo = [], 0
        for n in ctx.sizes:
            out.append(g[lo:lo + n])
            lo += n
        return (None, *out)


def joined_vector(forms: ParamForms, name: str, params: Sequence[torch.Tensor]) -> torch.Tensor:
    packed = forms.get(name, lambda: bias_form(list(params)), params[0].device)
    return _JoinedVector.apply(packed, *params)


ops.bind_all_functions(globals())     # every Function's backward runs in its forward's library context (ops.Context)
