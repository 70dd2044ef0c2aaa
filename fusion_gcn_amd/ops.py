"""Thin, checked Python wrappers over the C ABI (include/fgcn.h).

Every function takes torch CUDA tensors (float32, channels-last activations of shape (B, T, V, ld)), validates
what the kernels assume (device, dtype, contiguity, sizes) and enqueues on torch's current HIP stream.  Nothing
here computes anything in torch: a missing library or a non-gfx950 device raises ``FgcnError``.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GramItem, MixItem, MixTerm, TMap, check
from .paths import PathOptions, process_defaults

TMAP_POINTWISE = (1, 1, 0, 0, 1)


MATH_MODES = {"f32": 0, "bf16": 1, "bf16x3": 2, "f16x2": 2}     # FGCN_MATH_F32 / _BF16 / _BF16X3 (include/fgcn.h)
# "f16x2" = FGCN_MATH_BF16X3 with FGCN_PRODUCTS_F16X2 in the convolution / 1x1 kernels: the weight forms built in this mode are
# FGCN_PACK_SPLIT2H (``ScaledWeights``) and a kernel call takes the product form of the weights it is handed
X3_MODES = ("bf16x3", "f16x2")        # float32-accurate split modes (every kernel outside the conv / 1x1 family is bf16x3 in both)


class Context:
    """A set of library settings (math mode, product form, kernel-variant table: ``fgcn_ctx``, include/fgcn.h) that is current PER
    THREAD: ``with ops.context() as ctx:`` creates one from the thread's present settings, makes it current for this thread and
    restores the previous one on exit; ``ops.set_math_mode`` / ``fgcn_set_tuning`` inside change it and nothing else.  Two threads
    in two contexts (two models in two math modes on two streams) do not see each other's settings.  The autograd Functions of this
    package remember the context of their forward and make it current on the autograd thread for their backward (``context_bound``).
    Without any context a thread reads and writes the process-wide defaults (math mode bf16x3).
    ``paths``: which kernel form the blocks take for each stage (fusion_gcn_amd/paths.py) -- per context like the rest, so two models
    in one process may differ in them (tests/test_context_gpu.py runs two models with different PATH options on two threads)."""

    def __init__(self, handle: Optional[int], paths: Optional[PathOptions] = None):
        self.handle = handle            # None: the process-wide defaults
        self.f16x2 = False              # the math mode's product form ("f16x2" = FGCN_MATH_BF16X3 + two-way f16 products)
        self._paths = paths             # None: the process defaults (dataclass defaults + FGCN_PATHS), made on first use

    @property
    def paths(self) -> PathOptions:
        if self._paths is None:
            self._paths = process_defaults()
        return self._paths

    @paths.setter
    def paths(self, value: PathOptions) -> None:
        self._paths = value

    def __del__(self):                  # the library object goes with the last reference (a Function's ctx may outlive the `with` block)
        try:
            if self.handle is not None and _lib is not None:
                _lib.load().fgcn_ctx_destroy(self.handle)
        except Exception:               # noqa: BLE001 - interpreter shutdown
            pass
        self.handle = None


_DEFAULT_CONTEXT = Context(None)
_tls = threading.local()


def current_context() -> Context:
    return getattr(_tls, "ctx", None) or _DEFAULT_CONTEXT


@contextlib.contextmanager
def use_context(ctx: Optional[Context]):
    """Make ``ctx`` (None: the process-wide defaults) current on the calling thread for the duration."""
    ctx = ctx or _DEFAULT_CONTEXT
    prev = current_context()
    if ctx is prev:
        yield ctx
        return
    lib = _lib.load()
    check(lib.fgcn_ctx_set_current(ctx.handle), "fgcn_ctx_set_current")
    _tls.ctx = ctx
    try:
        yield ctx
    finally:
        check(lib.fgcn_ctx_set_current(prev.handle), "fgcn_ctx_set_current")
        _tls.ctx = prev


@contextlib.contextmanager
def context(mode: Optional[str] = None):
    """A fresh context (a copy of the thread's present settings, optionally with another math mode), current inside the block.  The
    object lives as long as anything refers to it: a forward recorded inside the block runs its backward in this context even after
    the block was left (``context_bound``)."""
    handle = ctypes.c_void_p()
    check(_lib.load().fgcn_ctx_create(ctypes.byref(handle)), "fgcn_ctx_create")
    ctx = Context(handle.value, current_context().paths.copy())
    ctx.f16x2 = current_context().f16x2
    with use_context(ctx):
        if mode is not None:
            set_math_mode(mode)
        yield ctx


def context_bound(fn_cls):
    """Class decorator for the package's torch.autograd.Function classes: the backward runs on an autograd worker thread, whose
    current context is not the forward's -- remember the forward's and make it current around the backward."""
    fwd, bwd = fn_cls.forward, fn_cls.backward

    def forward(ctx, *args, **kwargs):
        ctx._fgcn_context = current_context()
        return fwd(ctx, *args, **kwargs)

    def backward(ctx, *grads):
        # (a forward that bypassed this wrapper -- a subclass overriding forward without re-binding -- leaves no context: the
        # process-wide defaults then, rather than an AttributeError on the autograd thread)
        with use_context(getattr(ctx, "_fgcn_context", None)):
            return bwd(ctx, *grads)
    forward.__doc__, backward.__doc__ = fwd.__doc__, bwd.__doc__
    fn_cls.forward, fn_cls.backward = staticmethod(forward), staticmethod(backward)
    return fn_cls


def bind_all_functions(namespace: dict) -> None:
    """``context_bound`` for every autograd Function class defined in a module (call at the end of the module with ``globals()``)."""
    for obj in list(namespace.values()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function \
                and obj.__module__ == namespace.get("__name__") and "_fgcn_bound" not in obj.__dict__:
            # (the class's OWN attribute: a subclass of a bound Function that overrides forward / backward is bound again)
            context_bound(obj)
            obj._fgcn_bound = True


def set_math_mode(mode: str) -> None:
    """Arithmetic of the convolution / GEMM kernels in the calling thread's current context (the process-wide defaults when the
    thread has none): "bf16x3" (the default: float32-accurate -- operands split into three bfloat16 terms, six partial products per
    MFMA step, float32 accumulation; same tolerances as "f32") or "f32" (exact float32 MFMAs, the parity path) or "bf16" (BASELINE
    config 5: operands rounded to bfloat16 as the MFMA fragments are formed, float32 accumulation; everything in HBM, BatchNorm
    statistics, softmax and the joint mixing stay float32) or "f16x2" (float32-accurate as well: bf16x3 whose temporal / 1x1
    convolutions form every product from block-scaled two-way f16 splits, three MFMAs instead of six; include/fgcn.h
    FGCN_PRODUCTS_F16X2)."""
    if mode not in MATH_MODES:
        raise _lib.FgcnError(f"unknown math mode {mode!r} (f32 | bf16 | bf16x3 | f16x2)")
    check(_lib.load().fgcn_set_math_mode(MATH_MODES[mode]), "fgcn_set_math_mode")
    current_context().f16x2 = mode == "f16x2"
    check(_lib.load().fgcn_set_products(int(mode == "f16x2")), "fgcn_set_products")


def paths() -> PathOptions:
    """The calling thread's current path options (fusion_gcn_amd/paths.py)."""
    return current_context().paths


def get_math_mode() -> str:
    m = _lib.load().fgcn_get_math_mode()
    return "f16x2" if (m == 2 and current_context().f16x2) else {0: "f32", 1: "bf16", 2: "bf16x3"}[m]


class ScaledWeights:
    """A FGCN_PACK_SPLIT2H form (include/fgcn.h): one device buffer = 16-byte header (float bits of max |W|) + the two f16 parts of
    W * 2^s in the fragment order of the split kernels; what ``tconv_halo`` / ``pw_gemm`` stream with FGCN_PRODUCTS_F16X2."""

    def __init__(self, taps: int, K: int, N: int, device, acc_order: bool = False):
        self.taps, self.K, self.N, self.acc_order = taps, K, N, acc_order
        self.kgroups = (K + 15) // 16 * 2 if acc_order else (K + 7) // 8
        self.buf = torch.zeros(16 + 2 * taps * self.kgroups * N * 8 * 2, device=device, dtype=torch.uint8)

    device = property(lambda self: self.buf.device)

    def data_ptr(self) -> int:
        return self.buf.data_ptr()

    def amax(self) -> float:
        return float(self.buf[:4].view(torch.float32)[0])

    def parts(self) -> torch.Tensor:
        """(2, taps, K/8, N, 8) float16 view of the two parts (tests)."""
        return self.buf[16:].view(torch.float16).view(2, self.taps, self.kgroups, self.N, 8)


def _mode_products() -> None:
    """The product form of the current math mode (f16x2: two-way f16 splits) -- for entry points that take no weight form of their
    own to read it from, and after a wrapper that switched the form for one call (the library's product form is process-global)."""
    if _lib.load().fgcn_get_math_mode() == 2:
        check(_lib.load().fgcn_set_products(int(current_context().f16x2)), "fgcn_set_products")


def _use_products_of(w) -> None:
    """The conv / 1x1 kernels take the product form of the weights they are handed (in math mode bf16x3)."""
    if _lib.load().fgcn_get_math_mode() == 2:
        check(_lib.load().fgcn_set_products(int(isinstance(w, ScaledWeights))), "fgcn_set_products")


def pack_split2h(w: torch.Tensor) -> ScaledWeights:
    """(taps, K, N) f32 packed weights -> the FGCN_PACK_SPLIT2H form (two passes on the device: maximum, then the split)."""
    from .packing import Form, PackPlan, Seg
    ensure_device()
    _chk(w, "pack_split2h.w")
    taps, K, N = w.shape
    f = Form("split2h", taps, K, N, [Seg(w, st_tap=K * N, st_k=N, st_n=1, klen=K, nlen=N, tlen=taps)])
    f.alloc(w.device)
    PackPlan([f]).run()
    return f.dst


@contextlib.contextmanager
def math_mode(mode: str):
    """with ops.math_mode("bf16"): forward AND backward of the step -- the counterpart of the reference's
    MixedPrecisionStep (session/procedures/step.py:55-78, autocast around model(x)); no loss scaling is needed."""
    prev = get_math_mode()
    set_math_mode(mode)
    try:
        yield
    finally:
        set_math_mode(prev)


# weight-gradient kernels: stages (64 / 128 rows) a workgroup should at least walk before the rows are split further (small batches)
WGRAD_MIN_STAGES = 16


def conv_tmap(kt: int, stride: int) -> Tuple[int, int, int, int, int]:
    """Temporal map of a (kt x 1) convolution with padding (kt-1)//2 and the given stride."""
    return (kt, stride, 1, -((kt - 1) // 2), 1)


def conv_dgrad_tmap(kt: int, stride: int) -> Tuple[int, int, int, int, int]:
    """Map of that convolution's data gradient (output frames = the conv's input frames)."""
    return (kt, 1, -1, (kt - 1) // 2, stride)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.FgcnError(f"{name}: expected a contiguous float32 CUDA tensor, got {t.dtype} {t.device} "
                             f"contiguous={t.is_contiguous()}")


def _chk16(t: torch.Tensor, name: str) -> None:
    """a half-precision-storage tensor (include/fgcn.h, the `_h` entry points): contiguous bfloat16 on the device"""
    if not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous():
        raise _lib.FgcnError(f"{name}: expected a contiguous bfloat16 CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


def _chka(t: Optional[torch.Tensor], name: str) -> bool:
    """an activation tensor of a typed entry point (include/fgcn.h, `_t`): contiguous float32 or bfloat16 on the device -> whether it is
    bfloat16 (half-precision activation storage, math mode bf16); None -> False"""
    if t is None:
        return False
    if t.dtype == torch.bfloat16:
        _chk16(t, name)
        if get_math_mode() != "bf16":
            raise _lib.FgcnError(f"{name}: a bfloat16 activation tensor needs math mode bf16 (the mode is {get_math_mode()})")
        return True
    _chk(t, name)
    return False


def _half_mask(*flags: bool) -> int:
    """`half_mask` of a typed entry point: bit i = the i-th activation tensor (in the entry point's order) is bfloat16"""
    return sum(1 << i for i, f in enumerate(flags) if f)


def _p(t: Optional[torch.Tensor], coff: int = 0) -> Optional[int]:
    return None if t is None else t.data_ptr() + 4 * coff


_device_ok = False


def ensure_device() -> None:
    global _device_ok
    if not _device_ok:
        check(_lib.load().fgcn_check_device(), "fgcn_check_device")
        _device_ok = True


# ---- row GEMMs ---------------------------------------------------------------------------------------------------
def rows_gemm(inp: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, K: int, N: int, tmap=TMAP_POINTWISE,
              bias: Optional[torch.Tensor] = None, stats: bool = False, accumulate: bool = False,
              in_coff: int = 0, out_coff: int = 0) -> Optional[torch.Tensor]:
    """out[..., out_coff:out_coff+N] (+)= conv(inp[..., in_coff:in_coff+K]); w packed (taps, K, N).
    Returns the (tiles, 2, N) statistics partials when ``stats``.  Math mode bf16: ``inp`` / ``out`` may be bfloat16 tensors
    (half-precision activation storage, fgcn_rows_gemm_t; whole tensors, a bfloat16 ``out`` without accumulation)."""
    ensure_device()
    in16, out16 = _chka(inp, "rows_gemm.in"), _chka(out, "rows_gemm.out")
    _chk(w, "rows_gemm.w")
    if (in16 or out16) and (in_coff or out_coff or (out16 and accumulate)):
        raise _lib.FgcnError("rows_gemm: bfloat16 tensors are taken whole, a bfloat16 output without accumulation")
    B, T_in, V, ld_in = inp.shape
    Bo, T_out, Vo, ld_out = out.shape
    taps = tmap[0]
    if (Bo, Vo) != (B, V) or tuple(w.shape) != (taps, K, N):
        raise _lib.FgcnError(f"rows_gemm: shape mismatch in={tuple(inp.shape)} out={tuple(out.shape)} "
                             f"w={tuple(w.shape)} expected w=({taps},{K},{N})")
    if in_coff + K > ld_in or out_coff + N > ld_out or in_coff % 4 or out_coff % 4:
        raise _lib.FgcnError("rows_gemm: channel window outside the tensor or not 4-aligned")
    if bias is not None:
        _chk(bias, "rows_gemm.bias")
        if bias.numel() != N:
            raise _lib.FgcnError("rows_gemm: bias size")
    lib = _lib.load()
    part = None
    if stats:
        part = torch.empty((lib.fgcn_rows_gemm_tiles(B * T_out * V), 2, N), device=inp.device, dtype=torch.float32)
    if in16 or out16:
        check(lib.fgcn_rows_gemm_t(inp.data_ptr(), out.data_ptr(), _p(w), _p(bias), _p(part), B, T_in, T_out, V, K, N, ld_in, ld_out, TMap(*tmap),
                                   int(accumulate), _half_mask(in16, out16), _stream()), "fgcn_rows_gemm_t")
        return part
    check(lib.fgcn_rows_gemm(_p(inp, in_coff), _p(out, out_coff), _p(w), _p(bias), _p(part), B, T_in, T_out, V, K, N,
                             ld_in, ld_out, TMap(*tmap), int(accumulate), _stream()), "fgcn_rows_gemm")
    return part


def pack_k4(w: torch.Tensor) -> torch.Tensor:
    """(taps, K, N) packed weights -> k-interleaved (taps, K/4, N, 4) for tconv_halo (torch re-layout, tiny)."""
    taps, K, N = w.shape
    return w.view(taps, K // 4, 4, N).permute(0, 1, 3, 2).contiguous()


def pack_split3(w: torch.Tensor, acc_order: bool = False) -> torch.Tensor:
    """(taps, K, N) f32 packed weights -> the FGCN_MATH_BF16X3 form: (3, taps, ceil(K/8), N, 8) bfloat16, the exact
    three-way split w = w_h + w_m + w_l in the fragment order of v_mfma_f32_32x32x16_bf16.  ``acc_order``: the k order in
    which an MFMA accumulator enumerates its rows (spatial_fwd's weights; K is padded to a multiple of 16)."""
    ensure_device()
    _chk(w, "pack_split3.w")
    taps, K, N = w.shape
    k8 = (K + 15) // 16 * 2 if acc_order else (K + 7) // 8
    out = torch.empty((3, taps, k8, N, 8), device=w.device, dtype=torch.bfloat16)
    check(_lib.load().fgcn_pack_split3(_p(out), _p(w), taps, K, N, int(acc_order), _stream()), "fgcn_pack_split3")
    return out


SPLIT_MODES = ("bf16x3", "bf16", "f16x2")      # math modes whose halo-tile kernel streams split weights (bf16: part 0 only)


def pack_conv(w: torch.Tensor):
    """Packed (taps, K, N) weights in the form tconv_halo streams in the current math mode: ``pack_k4`` (f32),
    ``pack_split3`` (bf16x3, and bf16, which reads only the rounded high part) or ``pack_split2h`` (f16x2)."""
    mode = get_math_mode()
    if mode == "f16x2":
        return pack_split2h(w)
    return pack_split3(w) if mode in SPLIT_MODES else pack_k4(w)


def pack_spatial(wd: torch.Tensor, cin: int) -> torch.Tensor:
    """The stacked (K*Cin, Cout) conv_d matrix in the form spatial_fwd streams in the current math mode: ``pack_k4``
    (K*Cin/4, Cout, 4), or in bf16x3 (whole 32-channel tiles only) the accumulator-ordered three-way split."""
    if get_math_mode() == "f16x2" and cin % 32 == 0:
        from .packing import Form, PackPlan, Seg
        K, N = wd.shape
        f = Form("split2h_acc", 1, K, N, [Seg(wd, st_k=N, st_n=1, klen=K, nlen=N)])
        f.alloc(wd.device)
        PackPlan([f]).run()
        return f.dst
    if get_math_mode() in X3_MODES and cin % 32 == 0:
        return pack_split3(wd.unsqueeze(0), acc_order=True)
    return pack_k4(wd.unsqueeze(0))[0]


def tconv_halo_bn_sums() -> bool:
    """Whether ``tconv_halo(..., bn_bwd=...)`` is available in the current math mode (the split-bf16 kernel)."""
    return bool(_lib.load().fgcn_tconv_halo_bn_sums())


def tconv_halo(inp: torch.Tensor, w4: torch.Tensor, out: torch.Tensor, *, Th: int, taps: int, tb: int, tc: int,
               in_view=None, out_view=(1, 0), bias: Optional[torch.Tensor] = None, stats: bool = False,
               accumulate: bool = False, bn_bwd: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None,
               fuse_in: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]] = None,
               amax_out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Halo-tile temporal conv over virtual frames [0, Th): input frame th*in_s + in_o (th < Th_in), output frame
    th*out_s + out_o.  in_view = (in_s, in_o, Th_in); w4 from ``pack_k4``.  Returns stats partials when asked.
    ``bn_bwd = (a, sign image, vec)``: the call is the data gradient of a conv whose input was relu(BatchNorm(a) + shortcut); the
    returned partials (tiles, 2, N) then hold the BatchNorm-backward sums (sum dp, sum dp * a_hat) of what it writes
    (``bn_act_bwd(..., partials=)`` takes them instead of running its own reduction pass).
    ``amax_out`` (math mode f16x2): a zero-initialised one-element int32 tensor that receives the float bits of max |inp| over
    what the call stages (integer atomic maximum; several calls may share it) -- ``tconv_wgrad(amax=...)`` takes it.
    ``fuse_in = (vec, shortcut, g, g_sign)``: ``inp`` is the INPUT of a BatchNorm and the conv runs on
    g = relu(inp * scale + shift + shortcut), formed while the image is staged; g and its sign image (``bn_act``'s layout) are
    written as by-products -- the block's ``bn_act`` pass in front of the conv folded into it (split-bf16 kernel, taps > 1)."""
    ensure_device()
    in16 = inp.dtype == torch.bfloat16           # half-precision storage of the conv's input (math mode bf16: fgcn_tconv_halo_h)
    if in16:
        _chk16(inp, "tconv_halo.in")
        if get_math_mode() != "bf16" or fuse_in is not None or amax_out is not None:
            raise _lib.FgcnError("tconv_halo: a bfloat16 input needs math mode bf16 and takes neither a fused input stage nor amax_out")
    else:
        _chk(inp, "tconv_halo.in")
    out16 = _chka(out, "tconv_halo.out")         # half-precision ACTIVATION storage: the output as bfloat16 too (fgcn_tconv_halo_t)
    if out16 and (not in16 or accumulate or bn_bwd is not None):
        raise _lib.FgcnError("tconv_halo: a bfloat16 output comes with a bfloat16 input, without accumulation or BatchNorm-backward sums")
    B, T_in, V, ld_in = inp.shape
    Bo, T_out, Vo, ld_out = out.shape
    split = get_math_mode() in SPLIT_MODES       # the weights then are the pack_split3 form
    if isinstance(w4, ScaledWeights):
        if not split:
            raise _lib.FgcnError("tconv_halo: FGCN_PACK_SPLIT2H weights need math mode bf16x3 / f16x2")
        w_taps, K, N = w4.taps, w4.K, w4.N
    elif split:
        if w4.dtype != torch.bfloat16 or w4.dim() != 5 or w4.shape[0] != 3 or w4.shape[4] != 8 or not w4.is_contiguous():
            raise _lib.FgcnError(f"tconv_halo: math mode {get_math_mode()} takes pack_split3 weights, got {w4.dtype} {tuple(w4.shape)}")
        w_taps, K, N = w4.shape[1], w4.shape[2] * 8, w4.shape[3]
    else:
        _chk(w4, "tconv_halo.w4")
        w_taps, K, N = w4.shape[0], w4.shape[1] * 4, w4.shape[2]
    _use_products_of(w4)
    if (Bo, Vo) != (B, V) or w_taps != taps or (not split and w4.shape[3] != 4) or K > ld_in or N > ld_out:
        raise _lib.FgcnError(f"tconv_halo: shape mismatch in={tuple(inp.shape)} out={tuple(out.shape)} weights (taps, K, N) = {(w_taps, K, N)}")
    in_s, in_o, Th_in = in_view if in_view is not None else (1, 0, T_in)
    out_s, out_o = out_view
    lib = _lib.load()
    part = None
    bn = (None, None, None)
    if bn_bwd is not None:
        a, mask, vec = bn_bwd
        _chk(a, "tconv_halo.bn_a"), _chk(vec, "tconv_halo.bn_vec")
        if stats or tuple(a.shape) != tuple(out.shape) or mask.dtype != torch.uint8 or mask.numel() * 8 != out.numel() or vec.shape != (4, N):
            raise _lib.FgcnError("tconv_halo: bn_bwd needs a like out, its sign image and the (4, N) BatchNorm vector, and excludes stats")
        bn = (_p(a), mask.data_ptr(), _p(vec))
    if stats or bn_bwd is not None:
        part = torch.empty((lib.fgcn_tconv_halo_tiles(B, Th, Th_in, V), 2, N), device=inp.device, dtype=torch.float32)
    fin = (None, None, None, None)
    if fuse_in is not None:
        vec, res, g, g_sign = fuse_in
        _chk(vec, "tconv_halo.fin_vec"), _chk(res, "tconv_halo.fin_res"), _chk(g, "tconv_halo.fin_out")
        if tuple(res.shape) != tuple(inp.shape) or tuple(g.shape) != tuple(inp.shape) or vec.shape != (4, K) or ld_in != K or \
                g_sign.dtype != torch.uint8 or g_sign.numel() * 8 != inp.numel() or not g_sign.is_contiguous():
            raise _lib.FgcnError("tconv_halo: fuse_in needs the (4, K) BatchNorm vector, a shortcut and an output like the input "
                                 "(contiguous, K channels) and the uint8 sign image of numel / 8 bytes")
        fin = (_p(vec), _p(res), _p(g), g_sign.data_ptr())
    if out16:
        check(lib.fgcn_tconv_halo_t(inp.data_ptr(), out.data_ptr(), w4.data_ptr(), _p(bias), _p(part), B, Th, V, K, N, ld_in, ld_out,
                                    T_in, in_s, in_o, Th_in, T_out, out_s, out_o, taps, tb, tc, 3, _stream()), "fgcn_tconv_halo_t")
        return part
    if in16:
        check(lib.fgcn_tconv_halo_h(inp.data_ptr(), _p(out), w4.data_ptr(), _p(bias), _p(part), B, Th, V, K, N, ld_in, ld_out,
                                    T_in, in_s, in_o, Th_in, T_out, out_s, out_o, taps, tb, tc, int(accumulate), *bn, _stream()),
              "fgcn_tconv_halo_h")
        return part
    check(lib.fgcn_tconv_halo(_p(inp), _p(out), w4.data_ptr(), _p(bias), _p(part), B, Th, V, K, N, ld_in, ld_out,
                              T_in, in_s, in_o, Th_in, T_out, out_s, out_o, taps, tb, tc, int(accumulate), *bn, *fin,
                              None if amax_out is None else amax_out.data_ptr(), _stream()),
          "fgcn_tconv_halo")
    return part


def inference_kernels_available() -> bool:
    """Whether the inference forms of the two north-star kernels exist in the current math mode (the split kernels: bf16x3 / bf16)."""
    return get_math_mode() in ("bf16x3", "bf16")


def tconv_halo_bn_relu(inp: torch.Tensor, w4: torch.Tensor, out: torch.Tensor, *, taps: int, tb: int, tc: int, vec: torch.Tensor,
                       bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                       res_vec: Optional[torch.Tensor] = None) -> torch.Tensor:
    """North-star kernel 2 as stated, for inference: out = relu(BN(conv(inp) + bias) + [res | BN_res(res)]) in ONE kernel
    (fgcn_tconv_halo_bn_relu; stride 1; vec / res_vec = ``bn_eval_coeffs``; res laid out like out).  w4: the pack_split3 form."""
    ensure_device()
    _chk(inp, "tconv_halo_bn_relu.in"), _chk(out, "tconv_halo_bn_relu.out"), _chk(vec, "tconv_halo_bn_relu.vec")
    B, T, V, ld_in = inp.shape
    if w4.dtype != torch.bfloat16 or w4.dim() != 5 or w4.shape[0] != 3 or w4.shape[4] != 8 or not w4.is_contiguous():
        raise _lib.FgcnError(f"tconv_halo_bn_relu: pack_split3 weights expected, got {w4.dtype} {tuple(w4.shape)}")
    w_taps, K, N = w4.shape[1], w4.shape[2] * 8, w4.shape[3]
    if tuple(out.shape[:3]) != (B, T, V) or w_taps != taps or K > ld_in or N > out.shape[3] or tuple(vec.shape) != (4, N):
        raise _lib.FgcnError(f"tconv_halo_bn_relu: shape mismatch in={tuple(inp.shape)} out={tuple(out.shape)} weights (taps, K, N) = {(w_taps, K, N)}")
    if res is not None:
        _chk(res, "tconv_halo_bn_relu.res")
        if tuple(res.shape) != tuple(out.shape):
            raise _lib.FgcnError(f"tconv_halo_bn_relu: the shortcut {tuple(res.shape)} must be laid out like the output {tuple(out.shape)}")
    if res_vec is not None and (res is None or tuple(res_vec.shape) != (4, N)):
        raise _lib.FgcnError("tconv_halo_bn_relu: res_vec is the (4, N) BatchNorm vector of a given shortcut")
    _mode_products()
    check(_lib.load().fgcn_tconv_halo_bn_relu(_p(inp), _p(out), w4.data_ptr(), _p(bias), _p(vec), _p(res), _p(res_vec), B, T, V, K, N, ld_in,
                                              out.shape[3], taps, tb, tc, _stream()), "fgcn_tconv_halo_bn_relu")
    return out


def spatial_fwd_tile_bn_relu(x: torch.Tensor, a_hat: torch.Tensor, w3: torch.Tensor, bias: Optional[torch.Tensor], vec: torch.Tensor, *,
                             Cin: int, Cout: int, res: Optional[torch.Tensor] = None, res_vec: Optional[torch.Tensor] = None) -> torch.Tensor:
    """North-star kernel 1 for inference: g = relu(BN(sum_k conv_d[k](x . A^_k) + bias) + [res | BN_res(res)]) in ONE kernel
    (fgcn_spatial_fwd_tile_bn_relu; vec / res_vec = ``bn_eval_coeffs``; res (B, T, V, >= Cout))."""
    ensure_device()
    _chk(x, "spatial_fwd_tile_bn_relu.x"), _chk(a_hat, "spatial_fwd_tile_bn_relu.a_hat"), _chk(vec, "spatial_fwd_tile_bn_relu.vec")
    B, T, V, ld_x = x.shape
    if (w3.dtype != torch.bfloat16 or tuple(w3.shape) != (3, 1, 3 * Cin // 8, Cout, 8) or not w3.is_contiguous() or Cin > ld_x
            or a_hat.shape[0] not in (1, B) or tuple(a_hat.shape[1:]) != (3, V, V) or tuple(vec.shape) != (4, Cout)):
        raise _lib.FgcnError(f"spatial_fwd_tile_bn_relu: shape mismatch x={tuple(x.shape)} a_hat={tuple(a_hat.shape)} w3={tuple(w3.shape)} vec={tuple(vec.shape)}")
    ld_res = 0
    if res is not None:
        _chk(res, "spatial_fwd_tile_bn_relu.res")
        if tuple(res.shape[:3]) != (B, T, V) or res.shape[3] < Cout:
            raise _lib.FgcnError(f"spatial_fwd_tile_bn_relu: the shortcut {tuple(res.shape)} does not cover ({B}, {T}, {V}, {Cout})")
        ld_res = res.shape[3]
    if res_vec is not None and (res is None or tuple(res_vec.shape) != (4, Cout)):
        raise _lib.FgcnError("spatial_fwd_tile_bn_relu: res_vec is the (4, Cout) BatchNorm vector of a given shortcut")
    g = torch.empty((B, T, V, Cout), device=x.device, dtype=torch.float32)
    _mode_products()
    check(_lib.load().fgcn_spatial_fwd_tile_bn_relu(_p(x), _p(a_hat), w3.data_ptr(), _p(bias), _p(g), _p(vec), _p(res), ld_res, _p(res_vec),
                                                    B, T, V, Cin, Cout, ld_x, Cout, int(a_hat.shape[0] == B), _stream()),
          "fgcn_spatial_fwd_tile_bn_relu")
    return g


def pw_gemm_available() -> bool:
    """Whether ``pw_gemm`` runs in the current math mode (the split-bf16 modes)."""
    return bool(_lib.load().fgcn_pw_gemm_available())


def pw_gemm(inp: torch.Tensor, w3, out: torch.Tensor, *, bias: Optional[torch.Tensor] = None, stats: bool = False,
            accumulate: bool = False, amax_out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """1x1 convolution over all rows on the persistent split-bf16 row GEMM: inp (..., ld_in) and out (..., ld_out) contiguous with the
    same number of rows, w3 = pack_split3 of the (1, K, N) matrix.  Returns the BatchNorm partial sums (tiles, 2, N) when asked.
    Math mode bf16: ``inp`` / ``out`` may be bfloat16 tensors (fgcn_pw_gemm_t; a bfloat16 ``out`` without accumulation)."""
    ensure_device()
    in16, out16 = _chka(inp, "pw_gemm.in"), _chka(out, "pw_gemm.out")
    if out16 and accumulate:
        raise _lib.FgcnError("pw_gemm: a bfloat16 output is not accumulated into")
    if isinstance(w3, ScaledWeights):
        if w3.taps != 1:
            raise _lib.FgcnError("pw_gemm: a one-tap FGCN_PACK_SPLIT2H form expected")
        K, N = w3.K, w3.N
    elif w3.dtype != torch.bfloat16 or w3.dim() != 5 or w3.shape[0] != 3 or w3.shape[1] != 1 or w3.shape[4] != 8 or not w3.is_contiguous():
        raise _lib.FgcnError(f"pw_gemm: pack_split3 weights of a (1, K, N) matrix expected, got {w3.dtype} {tuple(w3.shape)}")
    else:
        K, N = w3.shape[2] * 8, w3.shape[3]
    _use_products_of(w3)
    ld_in, ld_out = inp.shape[-1], out.shape[-1]
    rows = inp.numel() // ld_in
    if out.numel() // ld_out != rows or K > ld_in or N > ld_out:
        raise _lib.FgcnError(f"pw_gemm: shape mismatch in={tuple(inp.shape)} out={tuple(out.shape)} weights (K, N) = {(K, N)}")
    lib = _lib.load()
    part = torch.empty((lib.fgcn_pw_gemm_tiles(rows), 2, N), device=inp.device, dtype=torch.float32) if stats else None
    if in16 or out16:
        check(lib.fgcn_pw_gemm_t(inp.data_ptr(), out.data_ptr(), w3.data_ptr(), _p(bias), _p(part), rows, K, N, ld_in, ld_out, int(accumulate),
                                 _half_mask(in16, out16), _stream()), "fgcn_pw_gemm_t")
        return part
    check(lib.fgcn_pw_gemm(_p(inp), _p(out), w3.data_ptr(), _p(bias), _p(part), rows, K, N, ld_in, ld_out, int(accumulate),
                           None if amax_out is None else amax_out.data_ptr(), _stream()), "fgcn_pw_gemm")
    return part


class ReduceBatch:
    """Leaf reductions of a block's backward, collected and issued as ONE fgcn_reduce_multi launch per 8 items.

    Inside ``with ops.deferred_reductions() as batch:`` the slab sums of the weight-gradient wrappers, and ``reduce_sum`` calls
    marked ``leaf=True``, only record (dst, src, shape); ``batch.flush()`` launches them on the current stream.  The caller
    must flush after every producer of the partial buffers has been enqueued on (or joined into) that stream, and must not
    read the results before.  The batch keeps the partial buffers referenced until ``release()``."""

    def __init__(self):
        self.items: list = []
        self.keep: list = []

    def add(self, dst: torch.Tensor, src: torch.Tensor, S: int, taps: int, K: int, N: int, K_dst: int, st_tap: int, st_k: int,
            st_n: int, accumulate: bool) -> None:
        self.items.append(_lib.ReduceItem(dst.data_ptr(), src.data_ptr(), st_tap, st_k, st_n, S, taps, K, N, K_dst, int(accumulate)))
        self.keep += [dst, src]

    def flush(self) -> None:
        lib = _lib.load()
        for lo in range(0, len(self.items), _lib.REDUCE_MAX_ITEMS):
            part = self.items[lo:lo + _lib.REDUCE_MAX_ITEMS]
            arr = (_lib.ReduceItem * len(part))(*part)
            check(lib.fgcn_reduce_multi(arr, len(part), _stream()), "fgcn_reduce_multi")
        self.items = []

    def release(self) -> None:
        self.keep = []


def _batch() -> Optional[ReduceBatch]:
    return getattr(_tls, "reduce_batch", None)


@contextlib.contextmanager
def deferred_reductions():
    """Collect leaf reductions instead of launching them one by one (see ReduceBatch); per thread, not re-entrant."""
    if _batch() is not None:
        raise _lib.FgcnError("deferred_reductions is not re-entrant")
    batch = _tls.reduce_batch = ReduceBatch()
    try:
        yield batch
    finally:
        _tls.reduce_batch = None
        if batch.items:          # an exception skipped the caller's flush: nothing may stay unreduced silently
            batch.flush()
        batch.release()


def _pick_nsplit(M: int, K: int, N: int, taps: int) -> int:
    tiles = ((K + 63) // 64) * ((N + 63) // 64) * taps
    want = max(1, 2048 // tiles)
    return int(max(1, min(want, (M + 511) // 512)))


def _reduce_slabs(partial: torch.Tensor, taps: int, K: int, N: int, out: Optional[torch.Tensor], accumulate: bool,
                  conv_param: Optional[Tuple[int, int]]) -> torch.Tensor:
    """Sum the (slabs, taps, K, N) partial weight gradients.  Default result: (taps, K, N).  ``conv_param=(groups, K_true)``
    writes the parameter layout instead: (groups, N, K_true, taps // groups, 1) -- `groups` stacked convolutions (conv_d's
    three subsets arrive as K = groups * K_in rows of one GEMM), input channels beyond K_true dropped (padding)."""
    slabs = partial.shape[0]
    if conv_param is None:
        if out is None:
            out = torch.empty((taps, K, N), device=partial.device, dtype=torch.float32)
        cur = _batch()
        batch = cur if cur is not None else ReduceBatch()
        batch.add(out, partial, slabs, taps, K, N, K, K * N, N, 1, accumulate)       # same kernel form as the parameter layout
        if batch is not cur:
            batch.flush()
        return out
    groups, k_true = conv_param
    if groups > 1:                      # (1, groups*K_in, N) reinterpreted as (groups, K_in, N): one "tap" per group
        if taps != 1 or K % groups:
            raise _lib.FgcnError("conv_param groups need a 1x1 convolution with K divisible by groups")
        g_taps, g_k, kt = groups, K // groups, 1
        shape, st_tap, st_k, st_n = (groups, N, k_true, 1, 1), N * k_true, 1, k_true
    else:
        g_taps, g_k, kt = taps, K, taps
        shape, st_tap, st_k, st_n = (N, k_true, kt, 1), 1, kt, k_true * kt
    if out is None:
        out = torch.empty(shape, device=partial.device, dtype=torch.float32)
    elif tuple(out.shape) != shape or not out.is_contiguous():
        raise _lib.FgcnError(f"weight-gradient output must be contiguous {shape}, got {tuple(out.shape)}")
    if _batch() is not None:
        _batch().add(out, partial, slabs, g_taps, g_k, N, k_true, st_tap, st_k, st_n, accumulate)
        return out
    one = ReduceBatch()          # on its own: the same kernel (its few-slabs form), one item
    one.add(out, partial, slabs, g_taps, g_k, N, k_true, st_tap, st_k, st_n, accumulate)
    one.flush()
    return out


def rows_wgrad(a: torch.Tensor, g: torch.Tensor, *, K: int, N: int, tmap=TMAP_POINTWISE, a_coff: int = 0,
               g_coff: int = 0, out: Optional[torch.Tensor] = None, accumulate: bool = False,
               wide: Optional[bool] = None, conv_param: Optional[Tuple[int, int]] = None,
               amax: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """(taps, K, N) weight gradient: sum over rows of a[src(row, tap), k] * g[row, n].  1x1 convolutions with K a multiple
    of 32 take the multi-accumulator kernel (``wide``; False forces the generic per-tap kernel).  ``conv_param`` returns
    the gradient in the convolution parameter's own layout (see ``_reduce_slabs``).  ``amax = (slot of a, slot of g)``: the
    one-element int32 tensors in which ``tconv_halo`` / ``pw_gemm`` left the operands' largest magnitudes (math mode f16x2: the
    split kernel then forms its products from two-way f16 splits; without them it runs bf16x3)."""
    ensure_device()
    a16, g16 = _chka(a, "rows_wgrad.a"), _chka(g, "rows_wgrad.g")      # math mode bf16: both bfloat16 (fgcn_pw_wgrad_h: the 1x1 form, K % 32 == 0)
    if a16 != g16 or (a16 and (a_coff or g_coff or amax is not None)):
        raise _lib.FgcnError("rows_wgrad: bfloat16 operands come as a pair, whole tensors")
    B, T_a, V, ld_a = a.shape
    Bg, T_g, Vg, ld_g = g.shape
    if (Bg, Vg) != (B, V) or a_coff + K > ld_a + 3 or g_coff + N > ld_g + 3 or a_coff % 4 or g_coff % 4:
        raise _lib.FgcnError(f"rows_wgrad: shape mismatch a={tuple(a.shape)} g={tuple(g.shape)} K={K} N={N}")
    taps = tmap[0]
    lib = _lib.load()
    ta, tb, tc, td = tmap[1:]
    if wide is None:
        # measured (tools/kbench.py wgrad): f32 +2-3 % at K = 384 / 768, -10..-25 % for narrower inputs; bf16 mode: faster or
        # equal at every width (the per-tap kernel's two LDS dwords per MFMA become the limit)
        wide = K >= 384 or get_math_mode() != "f32"
    if wide and taps == 1 and tc == 0 and td == 1 and ta >= 1 and K % 32 == 0 and (T_g - 1) * ta < T_a:
        # 1x1 (optionally strided) convolution: one accumulator per 32-channel chunk, every g fragment feeds 2-6 MFMAs
        chunks = lib.fgcn_pw_wgrad_chunks(K, N)
        tiles = ((K + 32 * chunks - 1) // (32 * chunks)) * ((N + 127) // 128 if N > 64 else 1)
        stages = B * ((T_g * V + 63) // 64) if N > 64 else B * ((T_g * V + 127) // 128)
        nsplit = max(1, min(lib.fgcn_pw_wgrad_resident(N) // max(tiles, 1), stages))   # every workgroup resident at once
        slabs = lib.fgcn_pw_wgrad_slabs(N, nsplit)
        partial = torch.empty((slabs, K, N), device=a.device, dtype=torch.float32)
        am = (None, None) if amax is None or a_coff or g_coff else (amax[0].data_ptr(), amax[1].data_ptr())
        if am[0] is not None and lib.fgcn_get_math_mode() == 2:
            check(lib.fgcn_set_products(1), "fgcn_set_products")   # (operand scales given: the f16x2 form of the split kernel)
        if a16:
            check(lib.fgcn_pw_wgrad_h(a.data_ptr(), g.data_ptr(), _p(partial), B, T_g, V, K, N, ld_a, ld_g, T_a, ta, 0, nsplit, _stream()),
                  "fgcn_pw_wgrad_h")
            return _reduce_slabs(partial.view(slabs, 1, K, N), 1, K, N, out, accumulate, conv_param)
        check(lib.fgcn_pw_wgrad(_p(a, a_coff), _p(g, g_coff), _p(partial), B, T_g, V, K, N, ld_a, ld_g, T_a, ta, 0,
                                nsplit, *am, _stream()), "fgcn_pw_wgrad")
        _mode_products()
        return _reduce_slabs(partial.view(slabs, 1, K, N), 1, K, N, out, accumulate, conv_param)
    if a16:
        raise _lib.FgcnError("rows_wgrad: bfloat16 operands take the 1x1 multi-accumulator form only (K % 32 == 0, one tap)")
    nsplit = _pick_nsplit(B * T_g * V, K, N, taps)
    partial = torch.empty((nsplit, taps, K, N), device=a.device, dtype=torch.float32)
    check(lib.fgcn_rows_wgrad(_p(a, a_coff), _p(g, g_coff), _p(partial), B, T_a, T_g, V, K, N, ld_a, ld_g,
                              TMap(*tmap), nsplit, _stream()), "fgcn_rows_wgrad")
    return _reduce_slabs(partial, taps, K, N, out, accumulate, conv_param)


TWGRAD_TAPS = (1, 2, 3, 4, 5, 6, 9)   # taps per call the multi-tap kernel is instantiated for (math mode f32)
TWGRAD_TAPS_SPLIT = (1, 2, 3, 4, 5, 9)   # ... in the split-bf16 modes (tap mode of tconv_wgrad_x3_kernel: fgcn_twgrad.hip)


def tconv_wgrad(a: torch.Tensor, g: torch.Tensor, *, taps: int, stride: int = 1, out: Optional[torch.Tensor] = None,
                accumulate: bool = False, all_taps: Optional[bool] = None,
                conv_param: Optional[Tuple[int, int]] = None,
                amax: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """(taps, K, N) weight gradient of the (taps x 1) temporal convolution with stride ``stride`` and padding
    (taps-1)//2: all taps in one pass over the rows (one call per residue class of the tap offset when strided).
    a: (B, T_a, V, K) conv input, g: (B, T_g, V, N) gradient of the conv output.  ``all_taps`` False forces the
    per-tap kernel (measured slower: 105-107 vs 107-126 TFLOP/s at 64-256 channels, 111-113 for the strided ones)."""
    ensure_device()
    in16 = a.dtype == torch.bfloat16 or g.dtype == torch.bfloat16     # half-precision storage of both operands (fgcn_tconv_wgrad_h)
    if in16:
        _chk16(a, "tconv_wgrad.a"), _chk16(g, "tconv_wgrad.g")
        if get_math_mode() != "bf16" or amax is not None or all_taps is False:
            raise _lib.FgcnError("tconv_wgrad: bfloat16 operands need math mode bf16 and the all-taps kernel")
    else:
        _chk(a, "tconv_wgrad.a"), _chk(g, "tconv_wgrad.g")
    B, T_a, V, K = a.shape
    Bg, T_g, Vg, N = g.shape
    pad = (taps - 1) // 2
    if (Bg, Vg) != (B, V) or T_g != (T_a - 1) // stride + 1:
        raise _lib.FgcnError(f"tconv_wgrad: shape mismatch a={tuple(a.shape)} g={tuple(g.shape)} stride={stride}")
    calls = []
    for par in range(stride):                       # taps j with (j - pad) % stride == par read frames of parity par
        js = [j for j in range(taps) if (j - pad) % stride == par]
        if js:
            calls.append((par, js[0], len(js), (js[0] - pad - par) // stride))
    if all_taps is None:
        all_taps = True
    ok_taps = TWGRAD_TAPS if get_math_mode() == "f32" else TWGRAD_TAPS_SPLIT
    if not all_taps or any(n not in ok_taps for _, _, n, _ in calls):
        if in16:
            raise _lib.FgcnError(f"tconv_wgrad: bfloat16 operands with {taps} taps, stride {stride}: no all-taps kernel for this pass")
        return rows_wgrad(a, g, K=K, N=N, tmap=conv_tmap(taps, stride), out=out, accumulate=accumulate, wide=False,
                          conv_param=conv_param)
    lib = _lib.load()
    tiles = ((K + 31) // 32) * ((N + 127) // 128 if N > 64 else 1)
    stages = B * ((T_g * V + 63) // 64) if N > 64 else B * ((T_g * V + 127) // 128)
    # at most 512 workgroups (two per CU, all resident at once; 256 for the one-workgroup-per-CU split-bf16 kernel); small
    # batches: >= 16 stages per workgroup while that still leaves a workgroup per CU -- fewer slabs for the reduction
    cap = max(1, lib.fgcn_tconv_wgrad_resident(N) // max(tiles, 1))
    nsplit = max(1, min(cap, max(stages // WGRAD_MIN_STAGES, min(stages, max(1, 256 // max(tiles, 1))))))
    slabs = lib.fgcn_tconv_wgrad_slabs(N, nsplit)
    partial = torch.empty((slabs, taps, K, N), device=a.device, dtype=torch.float32)
    for par, tap0, ntaps, shift0 in calls:
        th_a = (T_a - par + stride - 1) // stride
        if in16:
            check(lib.fgcn_tconv_wgrad_h(a.data_ptr(), g.data_ptr(), _p(partial), B, T_g, V, K, N, K, N, T_a, stride, par, th_a,
                                         ntaps, shift0, tap0, stride, taps, nsplit, _stream()), "fgcn_tconv_wgrad_h")
            continue
        am = (None, None) if amax is None else (amax[0].data_ptr(), amax[1].data_ptr())
        if am[0] is not None and lib.fgcn_get_math_mode() == 2:
            check(lib.fgcn_set_products(1), "fgcn_set_products")   # (operand scales given: the f16x2 form of the split kernel)
        check(lib.fgcn_tconv_wgrad(_p(a), _p(g), _p(partial), B, T_g, V, K, N, K, N, T_a, stride, par, th_a,
                                   ntaps, shift0, tap0, stride, taps, nsplit, *am, _stream()), "fgcn_tconv_wgrad")
    _mode_products()
    return _reduce_slabs(partial, taps, K, N, out, accumulate, conv_param)


def reduce_sum(src: torch.Tensor, dst: torch.Tensor, accumulate: bool = False, leaf: bool = False) -> torch.Tensor:
    """dst[i] (+)= sum_s src[s, i].  ``leaf``: nothing reads ``dst`` before the enclosing ``deferred_reductions`` batch is flushed,
    so the sum may ride in that batch's launch."""
    ensure_device()
    _chk(src, "reduce_sum.src"), _chk(dst, "reduce_sum.dst")
    S, count = src.shape[0], src[0].numel()
    if dst.numel() != count:
        raise _lib.FgcnError("reduce_sum: size mismatch")
    if leaf and _batch() is not None and count < (1 << 31):
        _batch().add(dst, src, S, 1, 1, count, 1, 0, 0, 1, accumulate)
        return dst
    check(_lib.load().fgcn_reduce_sum(_p(dst), _p(src), S, count, int(accumulate), _stream()), "fgcn_reduce_sum")
    return dst


def pack_weight(src: torch.Tensor, taps: int, K: int, N: int, st_tap: int, st_k: int, st_n: int,
                n_pad: Optional[int] = None, flip: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gather a conv weight into the packed (taps, K, N_pad) layout the row GEMM consumes."""
    ensure_device()
    _chk(src, "pack_weight.src")
    n_dst = N if n_pad is None else n_pad
    if out is None:
        out = torch.empty((taps, K, n_dst), device=src.device, dtype=torch.float32)
    check(_lib.load().fgcn_pack_weight(_p(out), _p(src), taps, K, N, n_dst, st_tap, st_k, st_n, int(flip), _stream()),
          "fgcn_pack_weight")
    return out


# ---- joint mixing --------------------------------------------------------------------------------------------------
def mix_items(spec: Sequence[dict]):
    """spec: [{out_c, width, terms: [(mat, transpose, in_c_lo, in_c_hi, mask), ...]}, ...] -> ctypes array."""
    arr = (MixItem * len(spec))()
    for i, it in enumerate(spec):
        arr[i].out_c, arr[i].width, arr[i].nterms = it["out_c"], it.get("width", 32), len(it["terms"])
        for j, (mat, tr, lo, hi, mask) in enumerate(it["terms"]):
            arr[i].term[j] = MixTerm(mat, tr, lo, hi, mask)
    return arr


def joint_mix(inp: torch.Tensor, out: torch.Tensor, mats: torch.Tensor, spec: Sequence[dict], *, in_channels: int,
              out_channels: int, accumulate: bool = False) -> torch.Tensor:
    """out[(n,t,u), oc] (+)= sum_terms sum_v M[u, v] in[(n,t,v), ic]; mats (B|1, n_mats, V, V)."""
    ensure_device()
    _chk(inp, "joint_mix.in"), _chk(out, "joint_mix.out"), _chk(mats, "joint_mix.mats")
    B, T, V, ld_in = inp.shape
    if out.shape[:3] != inp.shape[:3] or mats.shape[-1] != V or mats.shape[-2] != V or mats.shape[0] not in (1, B):
        raise _lib.FgcnError(f"joint_mix: shape mismatch in={tuple(inp.shape)} out={tuple(out.shape)} "
                             f"mats={tuple(mats.shape)}")
    items = mix_items(spec)
    check(_lib.load().fgcn_joint_mix(_p(inp), _p(out), _p(mats), B, T, V, ld_in, out.shape[3], in_channels,
                                     out_channels, mats.shape[1], int(mats.shape[0] != 1),
                                     items, len(spec), int(accumulate), _stream()), "fgcn_joint_mix")
    return out


MIX_MAX_ITEMS = 24   # FGCN_MIX_MAX_ITEMS (include/fgcn.h)


def joint_mix_vec(inp: torch.Tensor, out: torch.Tensor, mats: torch.Tensor, spec: Sequence[dict], *, vw: int,
                  accumulate: bool = False, colsum: bool = False, amax_out: Optional[torch.Tensor] = None):
    """Channel-group joint mix; spec: [{out_c, nch, terms: [(mat, transpose, in_c)]}].  ``colsum`` also returns the
    per-channel sums of everything written (ld_out,) -- a bias gradient without another pass over ``out``."""
    ensure_device()
    _chk(inp, "joint_mix_vec.in"), _chk(out, "joint_mix_vec.out"), _chk(mats, "joint_mix_vec.mats")
    B, T, V, ld_in = inp.shape
    if out.shape[:3] != inp.shape[:3] or mats.shape[-1] != V or mats.shape[-2] != V or mats.shape[0] not in (1, B):
        raise _lib.FgcnError(f"joint_mix_vec: shape mismatch in={tuple(inp.shape)} out={tuple(out.shape)} "
                             f"mats={tuple(mats.shape)}")
    lib = _lib.load()
    if colsum and (accumulate or len(spec) > MIX_MAX_ITEMS):
        raise _lib.FgcnError("joint_mix_vec: column sums need one launch without accumulation")
    partial = None
    if colsum:
        partial = torch.empty((B * lib.fgcn_joint_mix_chunks(B, T), out.shape[3]), device=inp.device, dtype=torch.float32)
    for lo in range(0, len(spec), MIX_MAX_ITEMS):   # one launch per FGCN_MIX_MAX_ITEMS items
        part = spec[lo:lo + MIX_MAX_ITEMS]
        arr = (_lib.MixVItem * len(part))()
        for i, it in enumerate(part):
            arr[i].out_c, arr[i].nch, arr[i].nterms = it["out_c"], it["nch"], len(it["terms"])
            for j, (mat, tr, in_c) in enumerate(it["terms"]):
                arr[i].term[j] = _lib.MixVTerm(mat, tr, in_c)
        check(lib.fgcn_joint_mix_vec(_p(inp), _p(out), _p(mats), B, T, V, ld_in, out.shape[3], mats.shape[1],
                                     int(mats.shape[0] != 1), arr, len(part), vw, int(accumulate), _p(partial),
                                     None if amax_out is None else amax_out.data_ptr(), _stream()), "fgcn_joint_mix_vec")
    if colsum:
        sums = torch.empty((out.shape[3],), device=inp.device, dtype=torch.float32)
        reduce_sum(partial, sums, leaf=True)          # a bias gradient: a leaf of the backward
        return out, sums
    return out


def gram_t_chunk(B: int, T: int) -> int:
    chunk = 32
    while chunk > 4 and B * ((T + chunk - 1) // chunk) < 1024:
        chunk //= 2
    return chunk


def joint_gram(in1: torch.Tensor, in2: torch.Tensor, items: Sequence[Tuple[int, int, int]]) -> torch.Tensor:
    """items: [(c1, c2, width)] (matrix i from item i) -> partial (B, nchunk, n, 32, 32)."""
    ensure_device()
    _chk(in1, "joint_gram.in1"), _chk(in2, "joint_gram.in2")
    B, T, V, ld1 = in1.shape
    if in2.shape[:3] != in1.shape[:3]:
        raise _lib.FgcnError("joint_gram: shape mismatch")
    n = len(items)
    arr = (GramItem * n)()
    for i, (c1, c2, width) in enumerate(items):
        arr[i] = GramItem(c1, c2, width, i)
    chunk = gram_t_chunk(B, T)
    nchunk = (T + chunk - 1) // chunk
    partial = torch.empty((B, nchunk, n, 32, 32), device=in1.device, dtype=torch.float32)
    check(_lib.load().fgcn_joint_gram(_p(in1), _p(in2), _p(partial), B, T, V, ld1, in2.shape[3], chunk, n, arr, n,
                                      _stream()), "fgcn_joint_gram")
    return partial


def spatial_wgrad(x: torch.Tensor, dy: torch.Tensor, mats: torch.Tensor, *, out: Optional[torch.Tensor] = None,
                  accumulate: bool = False, conv_param: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """conv_d weight gradient with agg = x . A^ recomputed on chip: (1, ns*Cin, Cout), or the parameter layout with
    ``conv_param=(ns, cin_true)``.  x (B,T,V,Cin), dy (B,T,V,Cout), mats (B or 1, ns, V, V)."""
    ensure_device()
    _chk(x, "spatial_wgrad.x"), _chk(dy, "spatial_wgrad.dy"), _chk(mats, "spatial_wgrad.mats")
    B, T, V, Cin = x.shape
    Cout, ns = dy.shape[3], mats.shape[1]
    if dy.shape[:3] != (B, T, V) or mats.shape[0] not in (1, B) or mats.shape[2:] != (V, V):
        raise _lib.FgcnError(f"spatial_wgrad: shape mismatch x={tuple(x.shape)} dy={tuple(dy.shape)} mats={tuple(mats.shape)}")
    lib = _lib.load()
    slabs = B * lib.fgcn_spatial_wgrad_chunks(B, T, Cin, Cout)
    partial = torch.empty((slabs, 1, ns * Cin, Cout), device=x.device, dtype=torch.float32)
    check(lib.fgcn_spatial_wgrad(_p(x), _p(dy), _p(mats), _p(partial), B, T, V, Cin, Cout, Cin, Cout, ns,
                                 int(mats.shape[0] != 1), _stream()), "fgcn_spatial_wgrad")
    return _reduce_slabs(partial, 1, ns * Cin, Cout, out, accumulate, conv_param)


def spatial_wgrad_tile_available(V: int, cin: int, cout: int) -> bool:
    return bool(_lib.load().fgcn_spatial_wgrad_tile_available(int(V), int(cin), int(cout)))


def spatial_wgrad_tile(x: torch.Tensor, dy: torch.Tensor, mats: torch.Tensor, *, out: Optional[torch.Tensor] = None,
                       accumulate: bool = False, conv_param: Optional[Tuple[int, int]] = None, cin: Optional[int] = None,
                       cout: Optional[int] = None) -> torch.Tensor:
    """``spatial_wgrad`` in tile form (fgcn_spatial_wgrad_tile: channels in 64s, split-bf16 math mode, three subsets): the same
    result layout.  x (B,T,V,ld_x), dy (B,T,V,ld_dy), mats (B or 1, 3, V, V); ``cin`` / ``cout``: the leading channels of wider
    rows that take part (default: all)."""
    ensure_device()
    dy16 = dy.dtype == torch.bfloat16            # half-precision storage of dy (math mode bf16: fgcn_spatial_wgrad_tile_h)
    if dy16:
        _chk16(dy, "spatial_wgrad_tile.dy")
        if get_math_mode() != "bf16":
            raise _lib.FgcnError("spatial_wgrad_tile: a bfloat16 dy needs math mode bf16")
    else:
        _chk(dy, "spatial_wgrad_tile.dy")
    x16 = _chka(x, "spatial_wgrad_tile.x")        # half-precision activation storage (fgcn_spatial_wgrad_tile_t): with a bfloat16 dy
    _chk(mats, "spatial_wgrad_tile.mats")
    if x16 and not dy16:
        raise _lib.FgcnError("spatial_wgrad_tile: a bfloat16 x comes with a bfloat16 dy")
    B, T, V, ld_x = x.shape
    ld_dy = dy.shape[3]
    Cin, Cout = ld_x if cin is None else int(cin), ld_dy if cout is None else int(cout)
    if not (0 < Cin <= ld_x and 0 < Cout <= ld_dy):
        raise _lib.FgcnError(f"spatial_wgrad_tile: cin={Cin} / cout={Cout} outside the rows ({ld_x}, {ld_dy})")
    if dy.shape[:3] != (B, T, V) or mats.shape[0] not in (1, B) or tuple(mats.shape[1:]) != (3, V, V):
        raise _lib.FgcnError(f"spatial_wgrad_tile: shape mismatch x={tuple(x.shape)} dy={tuple(dy.shape)} mats={tuple(mats.shape)}")
    lib = _lib.load()
    slabs = lib.fgcn_spatial_wgrad_tile_slabs(B, T, V, Cin, Cout)
    if slabs <= 0:
        raise _lib.FgcnError(f"spatial_wgrad_tile: sizes not supported: V={V} Cin={Cin} Cout={Cout}")
    partial = torch.empty((slabs, 1, 3 * Cin, Cout), device=x.device, dtype=torch.float32)
    if x16:
        check(lib.fgcn_spatial_wgrad_tile_t(x.data_ptr(), dy.data_ptr(), _p(mats), _p(partial), B, T, V, Cin, Cout, ld_x, ld_dy,
                                            int(mats.shape[0] != 1), 3, _stream()), "fgcn_spatial_wgrad_tile_t")
    elif dy16:
        check(lib.fgcn_spatial_wgrad_tile_h(_p(x), dy.data_ptr(), _p(mats), _p(partial), B, T, V, Cin, Cout, ld_x, ld_dy,
                                            int(mats.shape[0] != 1), _stream()), "fgcn_spatial_wgrad_tile_h")
    else:
        check(lib.fgcn_spatial_wgrad_tile(_p(x), _p(dy), _p(mats), _p(partial), B, T, V, Cin, Cout, ld_x, ld_dy,
                                          int(mats.shape[0] != 1), _stream()), "fgcn_spatial_wgrad_tile")
    return _reduce_slabs(partial, 1, 3 * Cin, Cout, out, accumulate, conv_param)


def joint_dagg(x: torch.Tensor, dagg: torch.Tensor, mats: torch.Tensor, dx: torch.Tensor, *, accumulate: bool,
               gated: Sequence[Tuple[torch.Tensor, torch.Tensor]] = ()) -> torch.Tensor:
    """dx (+)= sum_k dagg_k . A^_k^T and the partial grams dA^_k = x^T dagg_k in one pass over dagg.
    x (B,T,V,C), dagg (B,T,V,ns*C), mats (B or 1, ns, V, V), dx (B,T,V,>=C) -> partial (B, nchunk, ns, 32, 32).
    ``gated``: up to two (tensor, sign image) pairs added to dx where the image's bit is set (contiguous (B,T,V,C) tensors and
    bn_act's uint8 bit image of the same element count): the ReLU-gated gradients of the block's identity shortcuts."""
    ensure_device()
    if len(gated) > 2:
        raise _lib.FgcnError("joint_dagg: at most two gated addends")
    for e, m in gated:
        _chk(e, "joint_dagg.gated")
        if tuple(e.shape) != tuple(x.shape) or m.dtype != torch.uint8 or m.numel() * 8 != e.numel() or not m.is_cuda:
            raise _lib.FgcnError(f"joint_dagg: gated addend {tuple(e.shape)} / image {m.numel()} bytes do not match x {tuple(x.shape)}")
    ex = [(_p(e), m.data_ptr()) for e, m in gated] + [(None, None)] * (2 - len(gated))
    _chk(x, "joint_dagg.x"), _chk(dagg, "joint_dagg.dagg"), _chk(mats, "joint_dagg.mats"), _chk(dx, "joint_dagg.dx")
    B, T, V, C = x.shape
    ns = mats.shape[1]
    if dagg.shape != (B, T, V, ns * C) or dx.shape[:3] != (B, T, V) or mats.shape[0] not in (1, B) or mats.shape[2:] != (V, V):
        raise _lib.FgcnError(f"joint_dagg: shape mismatch x={tuple(x.shape)} dagg={tuple(dagg.shape)} mats={tuple(mats.shape)}")
    chunk = gram_t_chunk(B, T)
    nchunk = (T + chunk - 1) // chunk
    partial = torch.empty((B, nchunk, ns, 32, 32), device=x.device, dtype=torch.float32)
    check(_lib.load().fgcn_joint_dagg(_p(x), _p(dagg), _p(mats), _p(dx), _p(partial), B, T, V, C, C, ns * C, dx.shape[3], ns,
                                      int(mats.shape[0] != 1), chunk, int(accumulate), ex[0][0], ex[0][1], ex[1][0], ex[1][1],
                                      _stream()), "fgcn_joint_dagg")
    return partial


def adj_softmax_fwd(partial: Optional[torch.Tensor], scale: float, adj_a: torch.Tensor, B: int,
                    use_softmax: bool = True, adj_b: Optional[torch.Tensor] = None):
    """a_hat = softmax_v(scale * sum partial) + adj_a + adj_b  -> (C (B,K,V,V) or None, a_hat (B,K,V,V))."""
    ensure_device()
    _chk(adj_a, "adj_softmax_fwd.adj_a")
    if adj_b is not None:
        _chk(adj_b, "adj_softmax_fwd.adj_b")
    K, V, _ = adj_a.shape
    a_hat = torch.empty((B, K, V, V), device=adj_a.device, dtype=torch.float32)
    c_out = torch.empty_like(a_hat) if use_softmax else None
    nchunk = partial.shape[1] if partial is not None else 0
    check(_lib.load().fgcn_adj_softmax_fwd(_p(partial), nchunk, float(scale), _p(adj_a), _p(adj_b), _p(c_out), _p(a_hat), B,
                                           K, V, int(use_softmax), _stream()), "fgcn_adj_softmax_fwd")
    return c_out, a_hat


def adj_softmax_bwd(partial: torch.Tensor, scale: float, c_in: Optional[torch.Tensor], V: int):
    """-> (d_a_hat (B,K,V,V), dS (B,K,V,V) or None)."""
    ensure_device()
    _chk(partial, "adj_softmax_bwd.partial")
    B, nchunk, K = partial.shape[:3]
    d_a_hat = torch.empty((B, K, V, V), device=partial.device, dtype=torch.float32)
    d_s = torch.empty_like(d_a_hat) if c_in is not None else None
    check(_lib.load().fgcn_adj_softmax_bwd(_p(partial), nchunk, float(scale), _p(c_in), _p(d_a_hat), _p(d_s), B, K, V,
                                           _stream()), "fgcn_adj_softmax_bwd")
    return d_a_hat, d_s


# ---- BatchNorm / epilogues -------------------------------------------------------------------------------------------
def bn_finalize(partials: torch.Tensor, count: int, gamma, beta, running_mean=None, running_var=None,
                momentum: float = 0.1, eps: float = 1e-5) -> torch.Tensor:
    """-> vec (4, C) = mean, rstd, scale, shift; updates running stats in place when given."""
    ensure_device()
    _chk(partials, "bn_finalize.partials")
    C = partials.shape[-1]
    vec = torch.empty((4, C), device=partials.device, dtype=torch.float32)
    check(_lib.load().fgcn_bn_finalize(_p(partials), partials.shape[0], count, _p(gamma), _p(beta), _p(running_mean),
                                       _p(running_var), momentum, eps, _p(vec), C, _stream()), "fgcn_bn_finalize")
    return vec


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps: float = 1e-5) -> torch.Tensor:
    ensure_device()
    C = gamma.numel()
    vec = torch.empty((4, C), device=gamma.device, dtype=torch.float32)
    check(_lib.load().fgcn_bn_eval_coeffs(_p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(vec), C,
                                          _stream()), "fgcn_bn_eval_coeffs")
    return vec


def bn_act(a: torch.Tensor, vec_a: torch.Tensor, b: Optional[torch.Tensor] = None, vec_b: Optional[torch.Tensor] = None,
           relu: bool = True, out: Optional[torch.Tensor] = None, sign_mask: bool = False, out_bf16: bool = False):
    """act(a*scale_a + shift_a + [b | b*scale_b + shift_b]).  ``sign_mask``: -> (out, mask) where mask holds one bit per
    element, [out > 0] (uint8, numel/8; None when the element count is not a multiple of 8) -- what the backward's
    ReLU gate reads instead of ``out``.  ``out_bf16``: the result is stored as bfloat16 (round to nearest even; fgcn_bn_act_h) -- for a
    tensor that only the bf16 kernels' staging reads (math mode bf16: the temporal conv's input).  ``a`` / ``b`` may be bfloat16 tensors
    themselves (half-precision activation storage, math mode bf16: fgcn_bn_act_t)."""
    ensure_device()
    a16, b16 = _chka(a, "bn_act.a"), _chka(b, "bn_act.b")
    C = a.shape[-1]
    rows = a.numel() // C
    res_mode = 0 if b is None else (1 if vec_b is None else 2)
    if b is not None and b.shape != a.shape:
        raise _lib.FgcnError(f"bn_act: residual shape {tuple(b.shape)} != {tuple(a.shape)}")
    if out is None:
        out = torch.empty_like(a, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    mask = None
    if sign_mask and relu and a.numel() % 8 == 0:
        mask = torch.empty(a.numel() // 8, device=a.device, dtype=torch.uint8)
    if a16 or b16:
        (_chk16 if out_bf16 else _chk)(out, "bn_act.out")
        check(_lib.load().fgcn_bn_act_t(a.data_ptr(), _p(vec_a), None if b is None else b.data_ptr(), _p(vec_b), out.data_ptr(), _p(mask), rows, C,
                                        res_mode, int(relu), _half_mask(a16, b16, out_bf16), _stream()), "fgcn_bn_act_t")
    elif out_bf16:
        _chk16(out, "bn_act.out")
        check(_lib.load().fgcn_bn_act_h(_p(a), _p(vec_a), _p(b), _p(vec_b), out.data_ptr(), _p(mask), rows, C, res_mode, int(relu),
                                        _stream()), "fgcn_bn_act_h")
    else:
        check(_lib.load().fgcn_bn_act(_p(a), _p(vec_a), _p(b), _p(vec_b), _p(out), _p(mask), rows, C, res_mode, int(relu),
                                      _stream()), "fgcn_bn_act")
    return (out, mask) if sign_mask else out


def bn_act_pool(a: torch.Tensor, vec_a: torch.Tensor, b: Optional[torch.Tensor], vec_b: Optional[torch.Tensor], groups: int):
    """mean over every group's rows of relu(a*scale_a + shift_a + [b | b*scale_b + shift_b]) -> (pooled (groups, C), sign mask):
    ``bn_act`` + ``group_mean`` of the last block without the activation between them (fgcn_bn_act_pool).  a: (..., C) whose rows
    form ``groups`` equal consecutive groups; C % 8 == 0."""
    ensure_device()
    a16, b16 = _chka(a, "bn_act_pool.a"), _chka(b, "bn_act_pool.b")     # (bfloat16 operands: fgcn_bn_act_pool_t)
    C = a.shape[-1]
    rows = a.numel() // C
    if rows % groups or C % 8:
        raise _lib.FgcnError(f"bn_act_pool: {rows} rows in {groups} groups, C={C} (equal groups, C % 8 == 0)")
    if b is not None and b.shape != a.shape:
        raise _lib.FgcnError(f"bn_act_pool: residual shape {tuple(b.shape)} != {tuple(a.shape)}")
    res_mode = 0 if b is None else (1 if vec_b is None else 2)
    lib = _lib.load()
    splits = lib.fgcn_bn_act_pool_splits(groups, rows // groups)
    partial = torch.empty((groups * splits, C), device=a.device, dtype=torch.float32)
    pooled = torch.empty((groups, C), device=a.device, dtype=torch.float32)
    mask = torch.empty(a.numel() // 8, device=a.device, dtype=torch.uint8)
    if a16 or b16:
        check(lib.fgcn_bn_act_pool_t(a.data_ptr(), _p(vec_a), None if b is None else b.data_ptr(), _p(vec_b), _p(mask), _p(partial), _p(pooled),
                                     groups, rows // groups, C, res_mode, _half_mask(a16, b16), _stream()), "fgcn_bn_act_pool_t")
    else:
        check(lib.fgcn_bn_act_pool(_p(a), _p(vec_a), _p(b), _p(vec_b), _p(mask), _p(partial), _p(pooled), groups, rows // groups, C,
                                   res_mode, _stream()), "fgcn_bn_act_pool")
    return pooled, mask


def bn_act_bwd(dout: torch.Tensor, out: Optional[torch.Tensor], a: torch.Tensor, vec_a: torch.Tensor,
               b: Optional[torch.Tensor], vec_b: Optional[torch.Tensor], *, relu: bool = True, train: bool = True,
               res_mode: int, db: Optional[torch.Tensor] = None, db_accumulate: bool = False,
               sign_mask: Optional[torch.Tensor] = None, need_sums: bool = True, need_db: bool = True,
               partials: Optional[torch.Tensor] = None, grp_rows: int = 0, da_bf16: bool = False, db_bf16: bool = False):
    """Backward of bn_act.  Returns (da, db, sums (3, C)): sums[0] = d beta, sums[1] = d gamma_a, sums[2] = d gamma_b.
    ``da_bf16``: da is stored as bfloat16 (fgcn_bn_act_bwd_apply_h: the gradient of the temporal conv's output in math mode bf16).
    The ReLU gate is read from ``sign_mask`` (bn_act's bit image) when given, else from ``out``.  ``need_sums=False`` with
    ``train=False`` (no BatchNorm statistics in the graph: only the gate and the scale) skips the reduction pass.
    ``grp_rows`` > 0: ``dout`` is (rows / grp_rows, C), one row per group of consecutive rows -- the gradient of ``bn_act_pool``'s
    output, already divided by the group size -- instead of its rows x C broadcast.
    ``dout`` / ``a`` / ``b`` may be bfloat16 tensors (half-precision activation storage, math mode bf16: the `_t` entry points; the gate
    is then the sign image and a per-group ``dout`` stays float32); ``db_bf16``: a freshly allocated db (the shortcut branch's gradient) is
    bfloat16 too (typed operands, C % 8 == 0, no accumulation)."""
    ensure_device()
    d16, a16 = _chka(dout, "bn_act_bwd.dout"), _chka(a, "bn_act_bwd.a")
    b16 = _chka(b, "bn_act_bwd.b") if (b is not None and res_mode == 2) else False      # (an identity shortcut is not read)
    typed = d16 or a16 or b16
    if typed and ((relu and sign_mask is None) or (d16 and grp_rows)):
        raise _lib.FgcnError("bn_act_bwd: bfloat16 tensors need the sign image as the ReLU gate and a float32 per-group gradient")
    C = a.shape[-1]
    rows = a.numel() // C
    if grp_rows and (rows % grp_rows or tuple(dout.shape) != (rows // grp_rows, C) or partials is not None):
        raise _lib.FgcnError(f"bn_act_bwd: a per-group gradient for {rows} rows in groups of {grp_rows} is ({rows // max(grp_rows, 1)}, {C}), got {tuple(dout.shape)}")
    lib = _lib.load()
    if sign_mask is not None and (sign_mask.dtype != torch.uint8 or sign_mask.numel() * 8 != a.numel()):
        raise _lib.FgcnError("bn_act_bwd: sign_mask must be the uint8 bit image of bn_act (numel/8 bytes)")
    sums = None
    if partials is not None:            # the producer of dout already summed (tconv_halo bn_bwd): (tiles, 2, C) -> sums[0:2]
        if res_mode == 2 or partials.shape[1:] != (2, C):
            raise _lib.FgcnError("bn_act_bwd: precomputed partials are (tiles, 2, C) sums of a BatchNorm without a second BatchNorm branch")
        sums = torch.empty((3, C), device=a.device, dtype=torch.float32)
        reduce_sum(partials.view(partials.shape[0], -1), sums[:2].view(-1))
    elif need_sums or train:
        tiles = lib.fgcn_elem_tiles(rows)
        partials = torch.empty((tiles, 3, C), device=a.device, dtype=torch.float32)
        if typed:
            check(lib.fgcn_bn_act_bwd_reduce_t(dout.data_ptr(), grp_rows, None, _p(sign_mask), a.data_ptr(), _p(vec_a),
                                               None if b is None else b.data_ptr(), _p(vec_b), _p(partials), tiles, rows, C, res_mode, int(relu),
                                               _half_mask(d16, a16, b16), _stream()), "fgcn_bn_act_bwd_reduce_t")
        elif grp_rows:
            check(lib.fgcn_bn_act_bwd_reduce_g(_p(dout), grp_rows, _p(out), _p(sign_mask), _p(a), _p(vec_a), _p(b), _p(vec_b), _p(partials),
                                               tiles, rows, C, res_mode, int(relu), _stream()), "fgcn_bn_act_bwd_reduce_g")
        else:
            check(lib.fgcn_bn_act_bwd_reduce(_p(dout), _p(out), _p(sign_mask), _p(a), _p(vec_a), _p(b), _p(vec_b), _p(partials),
                                             tiles, rows, C, res_mode, int(relu), _stream()), "fgcn_bn_act_bwd_reduce")
        sums = torch.empty((3, C), device=a.device, dtype=torch.float32)
        reduce_sum(partials.view(tiles, -1), sums.view(-1))
    da = torch.empty_like(a, dtype=torch.bfloat16 if da_bf16 else torch.float32)
    db16 = False
    if res_mode != 0 and db is None and need_db:   # need_db=False (identity residual): the caller adds the gated gradient itself
        db16 = bool(db_bf16 and typed and C % 8 == 0)
        db = torch.empty_like(a, dtype=torch.bfloat16 if db16 else torch.float32)
    elif db is not None:
        _chk(db, "bn_act_bwd.db")
    if typed:
        check(lib.fgcn_bn_act_bwd_apply_t(dout.data_ptr(), grp_rows, None, _p(sign_mask), a.data_ptr(), _p(vec_a),
                                          None if b is None else b.data_ptr(), _p(vec_b), _p(sums), da.data_ptr(), None if db is None else db.data_ptr(), rows, C, res_mode,
                                          int(relu), int(train), int(db_accumulate), _half_mask(d16, a16, b16, da_bf16, db16), _stream()),
              "fgcn_bn_act_bwd_apply_t")
    elif da_bf16:
        check(lib.fgcn_bn_act_bwd_apply_h(_p(dout), grp_rows, _p(out), _p(sign_mask), _p(a), _p(vec_a), _p(b), _p(vec_b), _p(sums),
                                          da.data_ptr(), _p(db), rows, C, res_mode, int(relu), int(train), int(db_accumulate), _stream()),
              "fgcn_bn_act_bwd_apply_h")
    elif grp_rows:
        check(lib.fgcn_bn_act_bwd_apply_g(_p(dout), grp_rows, _p(out), _p(sign_mask), _p(a), _p(vec_a), _p(b), _p(vec_b), _p(sums), _p(da),
                                          _p(db), rows, C, res_mode, int(relu), int(train), int(db_accumulate), _stream()),
              "fgcn_bn_act_bwd_apply_g")
    else:
        check(lib.fgcn_bn_act_bwd_apply(_p(dout), _p(out), _p(sign_mask), _p(a), _p(vec_a), _p(b), _p(vec_b), _p(sums), _p(da),
                                        _p(db), rows, C, res_mode, int(relu), int(train), int(db_accumulate), _stream()),
              "fgcn_bn_act_bwd_apply")
    return da, db, sums


def bn_apply_window(a: torch.Tensor, vec_a: torch.Tensor, out: torch.Tensor, coff: int) -> None:
    """out[..., coff:coff + C] = a * scale + shift (vec_a = bn_finalize's / bn_eval_coeffs' vector): a plain BatchNorm whose result is a
    channel window of the wider contiguous tensor ``out`` (fgcn_bn_apply_ld)."""
    ensure_device()
    _chk(a, "bn_apply_window.a"), _chk(out, "bn_apply_window.out")
    C, ld = a.shape[-1], out.shape[-1]
    rows = a.numel() // C
    if out.numel() // ld != rows or coff < 0 or coff + C > ld or coff % 4:
        raise _lib.FgcnError(f"bn_apply_window: window [{coff}, {coff + C}) of {tuple(out.shape)} for {tuple(a.shape)}")
    check(_lib.load().fgcn_bn_apply_ld(_p(a), _p(vec_a), _p(out, coff), rows, C, ld, _stream()), "fgcn_bn_apply_ld")


def bn_bwd_window(dout: torch.Tensor, coff: int, a: torch.Tensor, vec_a: torch.Tensor, train: bool = True):
    """Backward of bn_apply_window from the gradient of the WIDE tensor: -> (da contiguous like ``a``, sums (3, C): [0] = d beta,
    [1] = d gamma); dout[..., coff:coff + C] is read in place (fgcn_bn_bwd_reduce_ld / _apply_ld)."""
    ensure_device()
    _chk(dout, "bn_bwd_window.dout"), _chk(a, "bn_bwd_window.a")
    C, ld = a.shape[-1], dout.shape[-1]
    rows = a.numel() // C
    if dout.numel() // ld != rows or coff < 0 or coff + C > ld or coff % 4:
        raise _lib.FgcnError(f"bn_bwd_window: window [{coff}, {coff + C}) of {tuple(dout.shape)} for {tuple(a.shape)}")
    lib = _lib.load()
    tiles = lib.fgcn_elem_tiles(rows)
    partials = torch.empty((tiles, 3, C), device=a.device, dtype=torch.float32)
    check(lib.fgcn_bn_bwd_reduce_ld(_p(dout, coff), ld, _p(a), _p(vec_a), _p(partials), tiles, rows, C, _stream()), "fgcn_bn_bwd_reduce_ld")
    sums = torch.empty((3, C), device=a.device, dtype=torch.float32)
    reduce_sum(partials.view(tiles, -1), sums.view(-1))
    da = torch.empty_like(a)
    check(lib.fgcn_bn_bwd_apply_ld(_p(dout, coff), ld, _p(a), _p(vec_a), _p(sums), _p(da), rows, C, int(train), _stream()), "fgcn_bn_bwd_apply_ld")
    return da, sums


def col_sum(x: torch.Tensor, C: int, coff: int = 0) -> torch.Tensor:
    """Per-channel sum over all rows of x[..., coff:coff+C] -> (C,)."""
    ensure_device()
    _chk(x, "col_sum.x")
    if C > 1024:            # the kernel's block covers up to 1024 channels: wider tensors in channel windows
        return torch.cat([col_sum(x, min(1024, C - c0), coff + c0) for c0 in range(0, C, 1024)])
    ld = x.shape[-1]
    rows = x.numel() // ld
    lib = _lib.load()
    tiles = lib.fgcn_elem_tiles(rows)
    partials = torch.empty((tiles, C), device=x.device, dtype=torch.float32)
    check(lib.fgcn_col_sum(_p(x, coff), _p(partials), rows, C, ld, _stream()), "fgcn_col_sum")
    out = torch.empty((C,), device=x.device, dtype=torch.float32)
    return reduce_sum(partials, out)


def group_mean(x: torch.Tensor) -> torch.Tensor:
    """(G, R, C) -> (G, C): mean over the R rows of every group (global average pooling), fixed summation order."""
    ensure_device()
    _chk(x, "group_mean.x")
    G, R, C = x.shape
    lib = _lib.load()
    splits = lib.fgcn_group_mean_splits(G, R)
    partial = torch.empty((G * splits, C), device=x.device, dtype=torch.float32)
    out = torch.empty((G, C), device=x.device, dtype=torch.float32)
    check(lib.fgcn_group_mean(_p(x), _p(partial), _p(out), G, R, C, C, _stream()), "fgcn_group_mean")
    return out


# ---- the two ends of the step: input BatchNorm and loss (fgcn_head.hip) ---------------------------------------------------------
def data_bn_stats(x: torch.Tensor) -> torch.Tensor:
    """x (N, M, T, V, C) -> BatchNorm partial sums (tiles, 2, M*V*C) of the (m, v, c) channels over (n, t)."""
    ensure_device()
    _chk(x, "data_bn_stats.x")
    N, M, T, V, C = x.shape
    lib = _lib.load()
    part = torch.empty((lib.fgcn_data_bn_tiles(N, T), 2, M * V * C), device=x.device, dtype=torch.float32)
    check(lib.fgcn_data_bn_stats(_p(x), _p(part), N, M, T, V, C, _stream()), "fgcn_data_bn_stats")
    return part


def data_bn_apply(x: torch.Tensor, vec: torch.Tensor, Cp: int) -> torch.Tensor:
    """-> (N*M, T, V, Cp) = x * scale + shift per (m, v, c) channel, pad channels zero."""
    ensure_device()
    _chk(x, "data_bn_apply.x"), _chk(vec, "data_bn_apply.vec")
    N, M, T, V, C = x.shape
    out = torch.empty((N * M, T, V, Cp), device=x.device, dtype=torch.float32)
    check(_lib.load().fgcn_data_bn_apply(_p(x), _p(vec), _p(out), N, M, T, V, C, Cp, _stream()), "fgcn_data_bn_apply")
    return out


def data_bn_bwd(dout: torch.Tensor, x: torch.Tensor, vec: torch.Tensor, train: bool, need_dx: bool):
    """dout (N*M, T, V, Cp), x (N, M, T, V, C) -> (d gamma, d beta, dx or None)."""
    ensure_device()
    _chk(dout, "data_bn_bwd.dout"), _chk(x, "data_bn_bwd.x")
    N, M, T, V, C = x.shape
    Cp = dout.shape[3]
    lib = _lib.load()
    part = torch.empty((lib.fgcn_data_bn_tiles(N, T), 2 * M * V * C), device=x.device, dtype=torch.float32)
    check(lib.fgcn_data_bn_bwd_reduce(_p(dout), _p(x), _p(vec), _p(part), N, M, T, V, C, Cp, _stream()), "fgcn_data_bn_bwd_reduce")
    sums = torch.empty((2, M * V * C), device=x.device, dtype=torch.float32)
    reduce_sum(part, sums.view(-1))
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        check(lib.fgcn_data_bn_bwd_apply(_p(dout), _p(x), _p(vec), _p(sums), _p(dx), N, M, T, V, C, Cp, int(train), _stream()),
              "fgcn_data_bn_bwd_apply")
    return sums[1], sums[0], dx


def cross_entropy_fwd(logits: torch.Tensor, labels: torch.Tensor):
    """logits (rows, classes) float32 with unit column stride (any row stride), labels int64 -> (loss (2,) = {mean, valid rows},
    probs (rows, classes))."""
    ensure_device()
    rows, classes = logits.shape
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.stride(1) == 1 and labels.dtype == torch.int64
            and labels.is_cuda and labels.is_contiguous() and labels.numel() == rows):
        raise _lib.FgcnError("cross_entropy: float32 logits (rows, classes) with contiguous classes and int64 labels (rows,) on the device")
    probs = torch.empty((rows, classes), device=logits.device, dtype=torch.float32)
    row_loss = torch.empty(rows, device=logits.device, dtype=torch.float32)
    loss = torch.empty(2, device=logits.device, dtype=torch.float32)
    check(_lib.load().fgcn_cross_entropy_fwd(logits.data_ptr(), labels.data_ptr(), _p(probs), _p(row_loss), _p(loss), rows, classes,
                                             logits.stride(0), _stream()), "fgcn_cross_entropy_fwd")
    return loss, probs


def cross_entropy_bwd(probs: torch.Tensor, labels: torch.Tensor, loss: torch.Tensor, dloss: torch.Tensor) -> torch.Tensor:
    ensure_device()
    rows, classes = probs.shape
    dl = torch.empty((rows, classes), device=probs.device, dtype=torch.float32)
    check(_lib.load().fgcn_cross_entropy_bwd(_p(probs), labels.data_ptr(), _p(loss), _p(dloss), _p(dl), rows, classes, classes,
                                             _stream()), "fgcn_cross_entropy_bwd")
    return dl


# ---- fused spatial forward -----------------------------------------------------------------------------------------
def spatial_fwd(x: torch.Tensor, a_hat: torch.Tensor, wd: torch.Tensor, bias_sum: Optional[torch.Tensor], *, Cin: int,
                Cout: int, stats: bool = True):
    """y = sum_k conv_d[k](x . A^_k) fused; wd = pack_k4 of the stacked (K*Cin, Cout) matrix, i.e. (K*Cin/4, Cout, 4).
    -> (y (B,T,V,Cout), stats partials or None)."""
    ensure_device()
    _chk(x, "spatial_fwd.x"), _chk(a_hat, "spatial_fwd.a_hat")
    B, T, V, ld_x = x.shape
    ns = a_hat.shape[1]
    if isinstance(wd, ScaledWeights):                      # FGCN_PACK_SPLIT2H_ACC: the f16x2 form of the kernel
        w_ok = get_math_mode() in X3_MODES and Cin % 32 == 0 and wd.acc_order and (wd.taps, wd.K, wd.N) == (1, ns * Cin, Cout)
    elif get_math_mode() in X3_MODES and Cin % 32 == 0:    # weights in the pack_spatial split form
        w_ok = wd.dtype == torch.bfloat16 and tuple(wd.shape) == (3, 1, ns * Cin // 8, Cout, 8) and wd.is_contiguous()
    else:
        w_ok = wd.dtype == torch.float32 and tuple(wd.shape) == (ns * Cin // 4, Cout, 4) and wd.is_contiguous()
    if not w_ok or a_hat.shape[0] not in (1, B) or a_hat.shape[2:] != (V, V):
        raise _lib.FgcnError(f"spatial_fwd: shape mismatch x={tuple(x.shape)} a_hat={tuple(a_hat.shape)} "
                             f"(math mode {get_math_mode()}: weights from pack_spatial)")
    _use_products_of(wd)
    lib = _lib.load()
    y = torch.empty((B, T, V, Cout), device=x.device, dtype=torch.float32)
    part = torch.empty((lib.fgcn_spatial_tiles(B, T), 2, Cout), device=x.device, dtype=torch.float32) if stats else None
    check(lib.fgcn_spatial_fwd(_p(x), _p(a_hat), wd.data_ptr(), _p(bias_sum), _p(y), _p(part), B, T, V, Cin, Cout, ld_x, Cout, ns,
                               int(a_hat.shape[0] == B), _stream()), "fgcn_spatial_fwd")
    return y, part


def spatial_fwd_tile_available(V: int, Cin: int, Cout: int) -> bool:
    """Whether ``spatial_fwd_tile`` runs these sizes in the current math mode (bf16x3 products or bf16, Cin % 64 == 0, 16 <= V <= 32).  A pure
    query: the answer is formed from the MODE's product form, whatever form the last kernel call of this context left selected."""
    if current_context().f16x2:
        return False
    return _lib.load().fgcn_get_math_mode() in (1, 2) and 16 <= V <= 32 and Cin % 64 == 0 and Cout % 4 == 0 and Cin > 0


def spatial_fwd_tile(x: torch.Tensor, a_hat: torch.Tensor, w3: torch.Tensor, bias_sum: Optional[torch.Tensor], *, Cin: int,
                     Cout: int, stats: bool = True, y_bf16: bool = False):
    """y = sum_k conv_d[k](x . A^_k), the tile form of the fused kernel (fgcn_spatial_tile.hip): w3 = ``pack_split3`` of the stacked
    (1, 3 Cin, Cout) matrix.  -> (y (B,T,V,Cout), stats partials or None).  Math mode bf16: ``x`` may be a bfloat16 tensor and ``y_bf16``
    stores y as bfloat16 (half-precision activation storage, fgcn_spatial_fwd_tile_t; the statistics are those of the float32 accumulators)."""
    ensure_device()
    x16 = _chka(x, "spatial_fwd_tile.x")
    _chk(a_hat, "spatial_fwd_tile.a_hat")
    if x16 and not y_bf16:
        raise _lib.FgcnError("spatial_fwd_tile: a bfloat16 x comes with a bfloat16 y")
    B, T, V, ld_x = x.shape
    if (w3.dtype != torch.bfloat16 or tuple(w3.shape) != (3, 1, 3 * Cin // 8, Cout, 8) or not w3.is_contiguous()
            or a_hat.shape[0] not in (1, B) or tuple(a_hat.shape[1:]) != (3, V, V)):
        raise _lib.FgcnError(f"spatial_fwd_tile: shape mismatch x={tuple(x.shape)} a_hat={tuple(a_hat.shape)} w3={tuple(w3.shape)} "
                             "(weights: pack_split3 of the (1, 3 Cin, Cout) matrix)")
    lib = _lib.load()
    _mode_products()
    y = torch.empty((B, T, V, Cout), device=x.device, dtype=torch.bfloat16 if y_bf16 else torch.float32)
    part = torch.empty((lib.fgcn_spatial_fwd_tile_tiles(B, T, V), 2, Cout), device=x.device, dtype=torch.float32) if stats else None
    if y_bf16:
        check(lib.fgcn_spatial_fwd_tile_t(x.data_ptr(), _p(a_hat), w3.data_ptr(), _p(bias_sum), y.data_ptr(), _p(part), B, T, V, Cin, Cout, ld_x,
                                          Cout, int(a_hat.shape[0] == B), _half_mask(x16, True), _stream()), "fgcn_spatial_fwd_tile_t")
        return y, part
    check(lib.fgcn_spatial_fwd_tile(_p(x), _p(a_hat), w3.data_ptr(), _p(bias_sum), _p(y), _p(part), B, T, V, Cin, Cout, ld_x, Cout,
                                    int(a_hat.shape[0] == B), _stream()), "fgcn_spatial_fwd_tile")
    return y, part


def spatial_bwd_tile_available(V: int, Cin: int, Cout: int) -> bool:
    """Whether ``spatial_bwd_tile`` runs these sizes in the current math mode (bf16x3 or f16x2: the kernel multiplies three-way bf16
    splits in both; Cin % 64 == 0, Cout % 64 == 0, 16 <= V <= 32)."""
    return bool(_lib.load().fgcn_spatial_bwd_tile_available(V, Cin, Cout))


def spatial_bwd_tile(dy: torch.Tensor, x: torch.Tensor, a_hat: torch.Tensor, w3: torch.Tensor, dx: torch.Tensor, *,
                     accumulate: bool, gated: Sequence[Tuple[torch.Tensor, torch.Tensor]] = ()) -> torch.Tensor:
    """The fused backward of the spatial stage (fgcn_spatial_bwd_tile.hip): dagg = dy . Wd stays on chip,
    dx (+)= sum_k dagg_k . A^_k^T, and the partial grams dA^_k = x^T dagg_k come back as (B, nseg, 3, 32, 32) -- joint_dagg's output
    format.  w3 = ``pack_split3`` of the (1, Cout, 3 Cin) matrix [o][k Cin + c] = Wd_k[o][c].  ``gated``: none or exactly two
    (tensor, sign image) pairs added to dx where the image's bit is set (as in ``joint_dagg``; not with ``accumulate``).  The FIRST pair
    may carry a third member, the number of consecutive samples per group: its tensor is then (B / group, Cin), one row per group, added
    to every row of the group's samples (the gradient of a pooled block output, ``bn_act_pool``).
    Math mode bf16, half-precision activation storage (fgcn_spatial_bwd_tile_t): x, dx and the gated addends bfloat16 TOGETHER (with a
    bfloat16 dy; a per-group first addend stays float32)."""
    ensure_device()
    if len(gated) not in (0, 2) or (gated and accumulate):
        raise _lib.FgcnError("spatial_bwd_tile: gated addends come as the pair of identity shortcuts, without accumulation")
    group = gated[0][2] if gated and len(gated[0]) == 3 else 0
    x16 = _chka(x, "spatial_bwd_tile.x")
    all16 = _chka(dx, "spatial_bwd_tile.dx")      # dx and the gated addends share a storage type; a bfloat16 dx comes with a bfloat16 x and dy
    for i, (e, m, *_) in enumerate(gated):
        if _chka(e, "spatial_bwd_tile.gated") != (all16 and not (i == 0 and group)):
            raise _lib.FgcnError("spatial_bwd_tile: the gated addends have dx's storage type (a per-group first addend: float32)")
        want = (x.shape[0] // group, x.shape[3]) if (i == 0 and group) else tuple(x.shape)
        if tuple(e.shape) != want or m.dtype != torch.uint8 or m.numel() * 8 != x.numel() or not m.is_cuda or (group and x.shape[0] % group):
            raise _lib.FgcnError(f"spatial_bwd_tile: gated addend {tuple(e.shape)} / image {m.numel()} bytes do not match x {tuple(x.shape)}")
    ex = [(e.data_ptr(), m.data_ptr()) for e, m, *_ in gated] + [(None, None)] * (2 - len(gated))
    dy16 = dy.dtype == torch.bfloat16            # half-precision storage of dy (math mode bf16: fgcn_spatial_bwd_tile_h)
    if dy16:
        _chk16(dy, "spatial_bwd_tile.dy")
        if get_math_mode() != "bf16":
            raise _lib.FgcnError("spatial_bwd_tile: a bfloat16 dy needs math mode bf16")
    else:
        _chk(dy, "spatial_bwd_tile.dy")
    _chk(a_hat, "spatial_bwd_tile.a_hat")
    if (all16 and not x16) or (x16 and not dy16):
        raise _lib.FgcnError("spatial_bwd_tile: a bfloat16 dx comes with a bfloat16 x, a bfloat16 x with a bfloat16 dy")
    B, T, V, Cin = x.shape
    Cout = dy.shape[3]
    if (w3.dtype != torch.bfloat16 or tuple(w3.shape) != (3, 1, Cout // 8, 3 * Cin, 8) or not w3.is_contiguous()
            or tuple(dy.shape[:3]) != (B, T, V) or tuple(dx.shape[:3]) != (B, T, V) or dx.shape[3] < Cin
            or a_hat.shape[0] not in (1, B) or tuple(a_hat.shape[1:]) != (3, V, V)):
        raise _lib.FgcnError(f"spatial_bwd_tile: shape mismatch dy={tuple(dy.shape)} x={tuple(x.shape)} a_hat={tuple(a_hat.shape)} "
                             f"dx={tuple(dx.shape)} w3={tuple(w3.shape)} (weights: pack_split3 of the (1, Cout, 3 Cin) matrix)")
    lib = _lib.load()
    _mode_products()
    nseg = lib.fgcn_spatial_bwd_tile_segments(B, T, V)
    partial = torch.empty((B, max(nseg, 1), 3, 32, 32), device=x.device, dtype=torch.float32)
    if x16:
        check(lib.fgcn_spatial_bwd_tile_t(dy.data_ptr(), x.data_ptr(), _p(a_hat), w3.data_ptr(), dx.data_ptr(), _p(partial), B, T, V, Cin, Cout, Cout,
                                          Cin, dx.shape[3], int(a_hat.shape[0] == B), int(accumulate), ex[0][0], group, ex[0][1], ex[1][0], ex[1][1],
                                          7 if all16 else 3, _stream()), "fgcn_spatial_bwd_tile_t")
        return partial
    if dy16:
        check(lib.fgcn_spatial_bwd_tile_h(dy.data_ptr(), _p(x), _p(a_hat), w3.data_ptr(), _p(dx), _p(partial), B, T, V, Cin, Cout, Cout, Cin,
                                          dx.shape[3], int(a_hat.shape[0] == B), int(accumulate), ex[0][0], group, ex[0][1], ex[1][0], ex[1][1],
                                          _stream()), "fgcn_spatial_bwd_tile_h")
        return partial
    if group:
        check(lib.fgcn_spatial_bwd_tile_g(_p(dy), _p(x), _p(a_hat), w3.data_ptr(), _p(dx), _p(partial), B, T, V, Cin, Cout, Cout, Cin,
                                          dx.shape[3], int(a_hat.shape[0] == B), ex[0][0], group, ex[0][1], ex[1][0], ex[1][1], _stream()),
              "fgcn_spatial_bwd_tile_g")
        return partial
    check(lib.fgcn_spatial_bwd_tile(_p(dy), _p(x), _p(a_hat), w3.data_ptr(), _p(dx), _p(partial), B, T, V, Cin, Cout, Cout, Cin,
                                    dx.shape[3], int(a_hat.shape[0] == B), int(accumulate), ex[0][0], ex[0][1], ex[1][0], ex[1][1], _stream()),
          "fgcn_spatial_bwd_tile")
    return partial


def emb_fwd_tile_available(V: int, ic: int, cin: int) -> bool:
    """Whether ``emb_fwd_tile`` runs these sizes in the current math mode (bf16x3, f16x2 or bf16; 16 <= V <= 32, ic 16 / 32 / 64, cin % 32 == 0)."""
    return bool(_lib.load().fgcn_emb_fwd_tile_available(int(V), int(ic), int(cin)))


def emb_fwd_tile(x: torch.Tensor, w3: torch.Tensor, bias: torch.Tensor, *, ic: int, cin: Optional[int] = None, write_emb: bool = True,
                 emb_bf16: bool = False):
    """-> (emb (B,T,V,6 ic) = x . Wemb + bias, partial (B, segments, 3, 32, 32) of the affinity grams theta_k^T phi_k) in one launch
    (fgcn_emb_fwd_tile.hip; agcn.py:104-106).  w3 = ``pack_split3`` of the (1, cin, 6 ic) matrix; ``partial`` goes to ``adj_softmax_fwd``.
    ``write_emb=False`` (inference: only the backward reads the embeddings): emb is not written and None comes back in its place.
    ``emb_bf16`` (math mode bf16): emb is stored as bfloat16 (fgcn_emb_fwd_tile_h; its readers ``emb_dx_tile`` / ``emb_wgrad_tile`` take it)."""
    ensure_device()
    x16 = _chka(x, "emb_fwd_tile.x")              # half-precision activation storage (math mode bf16: fgcn_emb_fwd_tile_t)
    _chk(bias, "emb_fwd_tile.bias")
    B, T, V, ld_x = x.shape
    cin = ld_x if cin is None else int(cin)
    if (w3.dtype != torch.bfloat16 or tuple(w3.shape) != (3, 1, cin // 8, 6 * ic, 8) or not w3.is_contiguous() or bias.numel() != 6 * ic
            or cin > ld_x):
        raise _lib.FgcnError(f"emb_fwd_tile: shape mismatch x={tuple(x.shape)} w3={tuple(w3.shape)} bias={tuple(bias.shape)} ic={ic} "
                             f"(weights: pack_split3 of the (1, cin, 6 ic) matrix)")
    lib = _lib.load()
    nseg = lib.fgcn_emb_fwd_tile_segments(B, T, V, ic)
    if nseg <= 0:
        raise _lib.FgcnError(f"emb_fwd_tile: sizes not supported: V={V} ic={ic}")
    partial = torch.empty((B, nseg, 3, 32, 32), device=x.device, dtype=torch.float32)
    if x16:
        e16 = bool(emb_bf16 and write_emb)
        emb = torch.empty((B, T, V, 6 * ic), device=x.device, dtype=torch.bfloat16 if e16 else torch.float32) if write_emb else None
        check(lib.fgcn_emb_fwd_tile_t(x.data_ptr(), w3.data_ptr(), _p(bias), None if emb is None else emb.data_ptr(), _p(partial), B, T, V, cin, ic,
                                      ld_x, 6 * ic, _half_mask(True, e16), _stream()), "fgcn_emb_fwd_tile_t")
        return emb, partial
    if emb_bf16 and write_emb:
        emb = torch.empty((B, T, V, 6 * ic), device=x.device, dtype=torch.bfloat16)
        check(lib.fgcn_emb_fwd_tile_h(_p(x), w3.data_ptr(), _p(bias), emb.data_ptr(), _p(partial), B, T, V, cin, ic, ld_x, 6 * ic, _stream()),
              "fgcn_emb_fwd_tile_h")
        return emb, partial
    emb = torch.empty((B, T, V, 6 * ic), device=x.device, dtype=torch.float32) if write_emb else None
    check(lib.fgcn_emb_fwd_tile(_p(x), w3.data_ptr(), _p(bias), _p(emb), _p(partial), B, T, V, cin, ic, ld_x, 6 * ic, _stream()),
          "fgcn_emb_fwd_tile")
    return emb, partial


def emb_tile_available(V: int, ic: int, cx: int) -> bool:
    """Whether ``emb_dx_tile`` / ``emb_wgrad_tile`` run these sizes in the current math mode (bf16x3, f16x2 -- the kernels multiply
    three-way bf16 splits in both -- or bf16; 16 <= V <= 32, ic % 16 == 0, cx % 64 == 0)."""
    return bool(_lib.load().fgcn_emb_tile_available(int(V), int(ic), int(cx)))


def _chk_emb(name: str, emb: torch.Tensor, d_s: torch.Tensor, ic: int) -> None:
    if emb.dtype == torch.bfloat16:       # half-precision storage (math mode bf16: the `_h` entry points)
        _chk16(emb, f"{name}.emb")
        if get_math_mode() != "bf16":
            raise _lib.FgcnError(f"{name}: a bfloat16 emb needs math mode bf16")
    else:
        _chk(emb, f"{name}.emb")
    _chk(d_s, f"{name}.d_s")
    B, T, V, ld_e = emb.shape
    if ld_e < 6 * ic or d_s.shape[0] not in (1, B) or tuple(d_s.shape[1:]) != (3, V, V):
        raise _lib.FgcnError(f"{name}: shape mismatch emb={tuple(emb.shape)} d_s={tuple(d_s.shape)} ic={ic}")


def emb_dx_tile(emb: torch.Tensor, d_s: torch.Tensor, w3: torch.Tensor, dx: torch.Tensor, *, ic: int, accumulate: bool,
                cx: Optional[int] = None, dx_old: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx (+)= demb . Wemb^T with the embedding gradient demb (d theta_k = dS_k . phi_k, d phi_k = dS_k^T . theta_k) formed on chip
    (fgcn_emb_tile.hip; backward of agcn.py:104-106).  emb (B,T,V,>=6 ic) = [th0 ph0 th1 ph1 th2 ph2], d_s (B or 1, 3, V, V),
    w3 = ``pack_split3`` of the (1, 6 ic, cx) matrix [j][c] = Wemb[j][c], dx (B,T,V,>=cx)."""
    ensure_device()
    _chk_emb("emb_dx_tile", emb, d_s, ic)
    dx16 = _chka(dx, "emb_dx_tile.dx")            # half-precision activation storage (fgcn_emb_dx_tile_t): with a bfloat16 emb
    if dx16 and emb.dtype != torch.bfloat16:
        raise _lib.FgcnError("emb_dx_tile: a bfloat16 dx comes with a bfloat16 emb")
    B, T, V, ld_e = emb.shape
    cx = dx.shape[3] if cx is None else int(cx)
    if (w3.dtype != torch.bfloat16 or tuple(w3.shape) != (3, 1, 6 * ic // 8, cx, 8) or not w3.is_contiguous()
            or tuple(dx.shape[:3]) != (B, T, V) or dx.shape[3] < cx):
        raise _lib.FgcnError(f"emb_dx_tile: shape mismatch emb={tuple(emb.shape)} dx={tuple(dx.shape)} w3={tuple(w3.shape)} "
                             f"(weights: pack_split3 of the (1, 6 ic, cx) matrix)")
    lib = _lib.load()
    batched = int(d_s.shape[0] != 1)
    ws = torch.empty(lib.fgcn_emb_dx_tile_workspace(B, batched), device=emb.device, dtype=torch.uint8)      # the split planes of dS, dS^T
    if dx_old is not None:      # dx = bfloat16(dx_old + term): the float32-accumulated gradient handed over as bfloat16 by its last writer
        _chk(dx_old, "emb_dx_tile.dx_old")
        if not (dx16 and accumulate) or tuple(dx_old.shape) != tuple(dx.shape):
            raise _lib.FgcnError("emb_dx_tile: dx_old (float32, dx's shape) comes with a bfloat16 dx and accumulate=True")
    if dx16:
        check(lib.fgcn_emb_dx_tile_t(emb.data_ptr(), _p(d_s), w3.data_ptr(), dx.data_ptr(), ws.data_ptr(), B, T, V, ic, cx, ld_e, dx.shape[3], batched,
                                     int(accumulate), _p(dx_old), 3, _stream()), "fgcn_emb_dx_tile_t")
        return dx
    if emb.dtype == torch.bfloat16:
        check(lib.fgcn_emb_dx_tile_h(emb.data_ptr(), _p(d_s), w3.data_ptr(), _p(dx), ws.data_ptr(), B, T, V, ic, cx, ld_e, dx.shape[3], batched,
                                     int(accumulate), _stream()), "fgcn_emb_dx_tile_h")
        return dx
    check(lib.fgcn_emb_dx_tile(_p(emb), _p(d_s), w3.data_ptr(), _p(dx), ws.data_ptr(), B, T, V, ic, cx, ld_e, dx.shape[3], batched,
                               int(accumulate), _stream()), "fgcn_emb_dx_tile")
    return dx


def emb_wgrad_tile(emb: torch.Tensor, x: torch.Tensor, d_s: torch.Tensor, *, ic: int, cx: Optional[int] = None):
    """-> (dWemb (6 ic, cx) = demb^T . x in the parameters' (out, in) order, dbemb (6 ic,) = column sums of demb), demb formed on chip
    (see ``emb_dx_tile``).  x (B,T,V,>=cx): the embedding convolutions' input."""
    ensure_device()
    _chk_emb("emb_wgrad_tile", emb, d_s, ic)
    x16 = _chka(x, "emb_wgrad_tile.x")            # half-precision activation storage (fgcn_emb_wgrad_tile_t): with a bfloat16 emb
    if x16 and emb.dtype != torch.bfloat16:
        raise _lib.FgcnError("emb_wgrad_tile: a bfloat16 x comes with a bfloat16 emb")
    B, T, V, ld_e = emb.shape
    cx = x.shape[3] if cx is None else int(cx)
    if tuple(x.shape[:3]) != (B, T, V) or x.shape[3] < cx:
        raise _lib.FgcnError(f"emb_wgrad_tile: shape mismatch emb={tuple(emb.shape)} x={tuple(x.shape)} cx={cx}")
    lib = _lib.load()
    slabs = lib.fgcn_emb_wgrad_tile_slabs(B, T, V, ic, cx)
    if slabs <= 0:
        raise _lib.FgcnError(f"emb_wgrad_tile: sizes not supported: V={V} ic={ic} cx={cx}")
    partial = torch.empty((slabs, 1, 6 * ic, cx), device=x.device, dtype=torch.float32)
    bpart = torch.empty((slabs, 6 * ic), device=x.device, dtype=torch.float32)
    if x16:
        check(lib.fgcn_emb_wgrad_tile_t(emb.data_ptr(), x.data_ptr(), _p(d_s), _p(partial), _p(bpart), B, T, V, ic, cx, ld_e, x.shape[3],
                                        int(d_s.shape[0] != 1), 3, _stream()), "fgcn_emb_wgrad_tile_t")
    elif emb.dtype == torch.bfloat16:
        check(lib.fgcn_emb_wgrad_tile_h(emb.data_ptr(), _p(x), _p(d_s), _p(partial), _p(bpart), B, T, V, ic, cx, ld_e, x.shape[3],
                                        int(d_s.shape[0] != 1), _stream()), "fgcn_emb_wgrad_tile_h")
    else:
        check(lib.fgcn_emb_wgrad_tile(_p(emb), _p(x), _p(d_s), _p(partial), _p(bpart), B, T, V, ic, cx, ld_e, x.shape[3],
                                      int(d_s.shape[0] != 1), _stream()), "fgcn_emb_wgrad_tile")
    gw = _reduce_slabs(partial, 1, 6 * ic, cx, None, False, None)[0]
    gb = torch.empty((6 * ic,), device=x.device, dtype=torch.float32)
    reduce_sum(bpart, gb, leaf=True)
    return gw, gb


def transpose(x: torch.Tensor, ld_out: Optional[int] = None) -> torch.Tensor:
    """(B, R, C) -> (B, C, ld_out) with out[b, c, r] = x[b, r, c] and the columns [R, ld_out) zero-filled (ld_out >= R)."""
    ensure_device()
    _chk(x, "transpose.x")
    B, R, C = x.shape
    ld_out = R if ld_out is None else ld_out
    out = torch.empty((B, C, ld_out), device=x.device, dtype=torch.float32)
    check(_lib.load().fgcn_transpose(_p(x), _p(out), B, R, C, C, ld_out, _stream()), "fgcn_transpose")
    return out


def transpose_into(x: torch.Tensor, rows: int) -> torch.Tensor:
    """(B, C, ld) feature-major with `rows` valid columns -> (B, rows, C): the inverse of ``transpose`` (drops the padding)."""
    ensure_device()
    _chk(x, "transpose_into.x")
    B, C, ld = x.shape
    out = torch.empty((B, rows, C), device=x.device, dtype=torch.float32)
    # out[b][r][c] = x[b][c][r]: a transpose of the (C x rows) image with input row stride ld
    check(_lib.load().fgcn_transpose(_p(x), _p(out), B, C, rows, ld, C, _stream()), "fgcn_transpose")
    return out


def row_softmax_fwd(st: torch.Tensor, adj_t: torch.Tensor, V: int, scale: float):
    """st (B, K, V, ld) transposed scores -> (c, a): c = softmax over the last axis of scale*st[..., :V], a = c + adj_t (K, V, ld);
    padding columns zero."""
    ensure_device()
    _chk(st, "row_softmax_fwd.st"), _chk(adj_t, "row_softmax_fwd.adj_t")
    B, K, Vr, ld = st.shape
    if Vr != V or tuple(adj_t.shape) != (K, V, ld):
        raise _lib.FgcnError(f"row_softmax_fwd: shape mismatch st={tuple(st.shape)} adj_t={tuple(adj_t.shape)}")
    c, a = torch.empty_like(st), torch.empty_like(st)
    check(_lib.load().fgcn_row_softmax_fwd(_p(st), _p(adj_t), _p(c), _p(a), B * K * V, V, ld, K * V, float(scale), _stream()),
          "fgcn_row_softmax_fwd")
    return c, a


def row_softmax_bwd(da: torch.Tensor, c: torch.Tensor, V: int, scale: float) -> torch.Tensor:
    ensure_device()
    _chk(da, "row_softmax_bwd.da"), _chk(c, "row_softmax_bwd.c")
    ds = torch.empty_like(da)
    rows = da.numel() // da.shape[-1]
    check(_lib.load().fgcn_row_softmax_bwd(_p(da), _p(c), _p(ds), rows, V, da.shape[-1], float(scale), _stream()),
          "fgcn_row_softmax_bwd")
    return ds


ROWS_GEMM_MAX_PROBLEMS = 65535      # grid.z of one fgcn_rows_gemm_batched / _batched2 launch


def rows_gemm_batched(inp: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, batch: int, rows: int, K: int, N: int,
                      ld_in: int, ld_out: int, in_bs: int, w_bs: int, out_bs: int, in_off: int = 0, w_off: int = 0,
                      out_off: int = 0, accumulate: bool = False, inner: int = 1, in_bs2: int = 0, w_bs2: int = 0,
                      out_bs2: int = 0) -> None:
    """``batch * inner`` independent products out_p (rows x N) (+)= in_p (rows x K) . w_p (K x N) in one launch; problem (b, i) starts
    at element ``*_off + b * *_bs + i * *_bs2`` of the (contiguous float32) tensors ``inp`` / ``w`` / ``out``."""
    ensure_device()
    _chk(inp, "rows_gemm_batched.in"), _chk(w, "rows_gemm_batched.w"), _chk(out, "rows_gemm_batched.out")
    last = lambda off, bs, bs2, span: off + (batch - 1) * bs + (inner - 1) * bs2 + span       # noqa: E731
    if (min(in_bs, w_bs, out_bs, in_bs2, w_bs2, out_bs2) < 0 or last(in_off, in_bs, in_bs2, (rows - 1) * ld_in + K) > inp.numel()
            or last(w_off, w_bs, w_bs2, K * N) > w.numel() or last(out_off, out_bs, out_bs2, (rows - 1) * ld_out + N) > out.numel()):
        raise _lib.FgcnError("rows_gemm_batched: a problem reaches outside its tensor")
    lib = _lib.load()
    # one launch carries batch * inner problems on grid.z (<= 65535): more outer problems go out in chunks
    step = max(1, ROWS_GEMM_MAX_PROBLEMS // inner)
    for b0 in range(0, batch, step):
        nb = min(step, batch - b0)
        i_off, o_off, k_off = in_off + b0 * in_bs, out_off + b0 * out_bs, w_off + b0 * w_bs
        if inner == 1:
            check(lib.fgcn_rows_gemm_batched(_p(inp, i_off), _p(out, o_off), _p(w, k_off), nb, in_bs, out_bs, w_bs, rows,
                                             K, N, ld_in, ld_out, int(accumulate), _stream()), "fgcn_rows_gemm_batched")
        else:
            check(lib.fgcn_rows_gemm_batched2(_p(inp, i_off), _p(out, o_off), _p(w, k_off), nb, in_bs, out_bs, w_bs, inner, in_bs2,
                                              out_bs2, w_bs2, rows, K, N, ld_in, ld_out, int(accumulate), _stream()),
                  "fgcn_rows_gemm_batched2")


# ---- MS-G3D data movement -------------------------------------------------------------------------------------------------------
def tmaxpool3_fwd(x: torch.Tensor, stride: int, coff: int = 0, C: Optional[int] = None):
    """(3 x 1) temporal max pooling (padding 1) of the channel window [coff, coff + C) of x (B, T, V, ld) -> (out (B, T', V, C),
    idx uint8 (B, T', V, C) = the winning tap)."""
    ensure_device()
    _chk(x, "tmaxpool3.x")
    B, T, V, ld = x.shape
    C = ld - coff if C is None else C
    To = (T - 1) // stride + 1
    out = torch.empty((B, To, V, C), device=x.device, dtype=torch.float32)
    idx = torch.empty((B, To, V, C), device=x.device, dtype=torch.uint8)
    check(_lib.load().fgcn_tmaxpool3_fwd(_p(x, coff), _p(out), idx.data_ptr(), B, T, To, V, C, ld, stride, _stream()), "fgcn_tmaxpool3_fwd")
    return out, idx


def tmaxpool3_bwd(dout: torch.Tensor, idx: torch.Tensor, T_in: int, stride: int, din: Optional[torch.Tensor] = None,
                  coff: int = 0, accumulate: bool = False) -> torch.Tensor:
    """Gradient of tmaxpool3_fwd into (the channel window at ``coff`` of) din (B, T_in, V, ld); a fresh (B, T_in, V, C) when None."""
    ensure_device()
    _chk(dout, "tmaxpool3_bwd.dout")
    B, To, V, C = dout.shape
    if din is None:
        din = torch.empty((B, T_in, V, C), device=dout.device, dtype=torch.float32)
    _chk(din, "tmaxpool3_bwd.din")
    check(_lib.load().fgcn_tmaxpool3_bwd(_p(dout), idx.data_ptr(), _p(din, coff), B, T_in, To, V, C, din.shape[3], stride,
                                         int(accumulate), _stream()), "fgcn_tmaxpool3_bwd")
    return din


def unfold_out_frames(T: int, window: int, stride: int, dilation: int) -> int:
    pad = (window + (window - 1) * (dilation - 1) - 1) // 2
    return (T + 2 * pad - dilation * (window - 1) - 1) // stride + 1


def unfold_windows(x: torch.Tensor, window: int, stride: int, dilation: int = 1) -> torch.Tensor:
    """x (B, T, V, C) -> (B, T', window * V, C): the frames around every stride-th frame as one node axis."""
    ensure_device()
    _chk(x, "unfold_windows.x")
    B, T, V, C = x.shape
    To = unfold_out_frames(T, window, stride, dilation)
    out = torch.empty((B, To, window * V, C), device=x.device, dtype=torch.float32)
    check(_lib.load().fgcn_unfold_windows(_p(x), _p(out), B, T, To, V, C, window, stride, dilation, 0, _stream()), "fgcn_unfold_windows")
    return out


def unfold_windows_bwd(dout: torch.Tensor, T: int, V: int, window: int, stride: int, dilation: int = 1) -> torch.Tensor:
    ensure_device()
    _chk(dout, "unfold_windows_bwd.dout")
    B, To, WV, C = dout.shape
    dx = torch.empty((B, T, V, C), device=dout.device, dtype=torch.float32)
    check(_lib.load().fgcn_unfold_windows(_p(dout), _p(dx), B, T, To, V, C, window, stride, dilation, 1, _stream()), "fgcn_unfold_windows")
    return dx


_identity_vecs = {}


def col_moments(x: torch.Tensor) -> torch.Tensor:
    """BatchNorm partial sums of a tensor no GEMM epilogue produced: (tiles, 2, C) = per-tile (sum x, sum x*x) per channel, the
    layout ``bn_finalize`` takes.  Runs the BatchNorm-backward reduction kernel with dout = a = x and identity statistics
    (sum dout = sum x, sum dout * (a - 0) * 1 = sum x*x)."""
    ensure_device()
    _chk(x, "col_moments.x")
    C = x.shape[-1]
    rows = x.numel() // C
    lib = _lib.load()
    tiles = lib.fgcn_elem_tiles(rows)
    key = (C, str(x.device))
    vec = _identity_vecs.get(key)
    if vec is None:                      # {mean 0, rstd 1, scale 1, shift 0}: read-only, cached
        vec = torch.zeros((4, C), device=x.device, dtype=torch.float32)
        vec[1:3] = 1.0
        _identity_vecs[key] = vec
    partials = torch.empty((tiles, 3, C), device=x.device, dtype=torch.float32)
    check(lib.fgcn_bn_act_bwd_reduce(_p(x), None, None, _p(x), _p(vec), None, None, _p(partials), tiles, rows, C, 0, 0, _stream()),
          "fgcn_bn_act_bwd_reduce")
    return partials[:, :2].contiguous()
