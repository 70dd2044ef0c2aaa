"""ctypes binding of libfgcn.so (the C ABI declared in include/fgcn.h).

No fallback: if the library is missing or the device is not gfx950 every entry point raises.  Build it with
``python -m fusion_gcn_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FGCN_LIB") or os.path.join(_HERE, "libfgcn.so")   # FGCN_LIB: timing-probe builds (tools/probes)

c_float_p = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class TMap(C.Structure):
    """fgcn_tmap: ti = (to*ta + j*tb + tc) / td."""
    _fields_ = [("taps", C.c_int), ("ta", C.c_int), ("tb", C.c_int), ("tc", C.c_int), ("td", C.c_int)]


class MixTerm(C.Structure):
    _fields_ = [("mat", C.c_short), ("transpose", C.c_short), ("in_c_lo", C.c_short), ("in_c_hi", C.c_short),
                ("mask", C.c_short)]


class MixItem(C.Structure):
    _fields_ = [("out_c", C.c_short), ("width", C.c_short), ("nterms", C.c_short), ("term", MixTerm * 3)]


class MixVTerm(C.Structure):
    _fields_ = [("mat", C.c_short), ("transpose", C.c_short), ("in_c", C.c_short)]


class MixVItem(C.Structure):
    _fields_ = [("out_c", C.c_short), ("nch", C.c_short), ("nterms", C.c_short), ("term", MixVTerm * 3)]


class GramItem(C.Structure):
    _fields_ = [("c1", C.c_short), ("c2", C.c_short), ("width", C.c_short), ("mat", C.c_short)]


class PackSeg(C.Structure):
    """fgcn_pack_seg: one strided window of a parameter tensor inside a packed form."""
    _fields_ = [("src", C.c_void_p), ("st_tap", C.c_longlong), ("st_k", C.c_longlong), ("st_n", C.c_longlong),
                ("t0", C.c_int), ("tlen", C.c_int), ("k0", C.c_int), ("klen", C.c_int), ("n0", C.c_int), ("nlen", C.c_int),
                ("tap0", C.c_int), ("tap_step", C.c_int)]


PACK_MAX_SEG = 6            # FGCN_PACK_MAX_SEG
PACK_MODES = {"plain": 0, "k4": 1, "split3": 2, "split3_acc": 3, "split2h": 4, "split2h_acc": 5}       # FGCN_PACK_*


class PackItem(C.Structure):
    """fgcn_pack_item: one packed / split weight form = a sum of segments + the layout its consumer streams."""
    _fields_ = [("dst", C.c_void_p), ("mode", C.c_int), ("taps", C.c_int), ("K", C.c_int), ("N", C.c_int),
                ("kgroups", C.c_int), ("nseg", C.c_int), ("seg", PackSeg * PACK_MAX_SEG)]


class ReduceItem(C.Structure):
    """fgcn_reduce_item: one slab sum of a batched fgcn_reduce_multi launch."""
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("st_tap", C.c_longlong), ("st_k", C.c_longlong), ("st_n", C.c_longlong),
                ("S", C.c_int), ("taps", C.c_int), ("K", C.c_int), ("N", C.c_int), ("K_dst", C.c_int), ("accumulate", C.c_int)]


REDUCE_MAX_ITEMS = 8        # FGCN_REDUCE_MAX_ITEMS

_I, _LL, _F, _P = C.c_int, C.c_longlong, C.c_float, C.c_void_p

# name -> (restype, argtypes); mirrors include/fgcn.h one to one
SIGNATURES = {
    "fgcn_version": (_I, []),
    "fgcn_last_error": (C.c_char_p, []),
    "fgcn_check_device": (_I, []),
    "fgcn_ctx_create": (_I, [C.POINTER(C.c_void_p)]),
    "fgcn_ctx_destroy": (_I, [_P]),
    "fgcn_ctx_set_current": (_I, [_P]),
    "fgcn_ctx_get_current": (_P, []),
    "fgcn_set_tuning": (_I, [_I, _I]),
    "fgcn_get_tuning": (_I, [_I]),
    "fgcn_set_math_mode": (_I, [_I]),
    "fgcn_get_math_mode": (_I, []),
    "fgcn_set_products": (_I, [_I]),
    "fgcn_get_products": (_I, []),
    "fgcn_rows_gemm": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, TMap, _I, _P]),
    "fgcn_rows_gemm_batched": (_I, [_P, _P, _P, _I, _LL, _LL, _LL, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_rows_gemm_batched2": (_I, [_P, _P, _P, _I, _LL, _LL, _LL, _I, _LL, _LL, _LL, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_rows_gemm_tiles": (_I, [_LL]),
    "fgcn_tconv_halo_tiles": (_I, [_I, _I, _I, _I]),
    "fgcn_tconv_halo": (_I, [_P, _P, _P, _P, _P] + [_I] * 18 + [_P, _P, _P] + [_P, _P, _P, _P] + [_P, _P]),
    "fgcn_tconv_halo_bn_sums": (_I, []),
    "fgcn_rows_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, TMap, _I, _P]),
    "fgcn_tconv_wgrad_slabs": (_I, [_I, _I]),
    "fgcn_pw_wgrad_slabs": (_I, [_I, _I]),
    "fgcn_tconv_wgrad_resident": (_I, [_I]),
    "fgcn_pw_wgrad_resident": (_I, [_I]),
    "fgcn_tconv_wgrad": (_I, [_P, _P, _P] + [_I] * 17 + [_P, _P, _P]),
    "fgcn_pw_wgrad_chunks": (_I, [_I, _I]),
    "fgcn_pw_wgrad": (_I, [_P, _P, _P] + [_I] * 11 + [_P, _P, _P]),
    "fgcn_reduce_sum": (_I, [_P, _P, _I, _LL, _I, _P]),
    "fgcn_reduce_sum_strided": (_I, [_P, _P, _I, _I, _I, _I, _I, _LL, _LL, _LL, _I, _P]),
    "fgcn_reduce_multi": (_I, [C.POINTER(ReduceItem), _I, _P]),
    "fgcn_group_mean_splits": (_I, [_I, _I]),
    "fgcn_group_mean": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "fgcn_pack_split3": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fgcn_pack_weight": (_I, [_P, _P, _I, _I, _I, _I, _LL, _LL, _LL, _I, _P]),
    "fgcn_pack_kgroups": (_I, [_I, _I]),
    "fgcn_pack_units": (_LL, [_I, _I, _I, _I]),
    "fgcn_pack_run": (_I, [_P, _P, _I, _P]),
    "fgcn_pack_run_scaled": (_I, [_P, _P, _I, _I, _P]),
    "fgcn_joint_mix": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(MixItem), _I, _I, _P]),
    "fgcn_joint_mix_vec": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(MixVItem), _I, _I, _I, _P, _P, _P]),
    "fgcn_joint_mix_chunks": (_I, [_I, _I]),
    "fgcn_joint_gram": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(GramItem), _I, _P]),
    "fgcn_spatial_wgrad_chunks": (_I, [_I, _I, _I, _I]),
    "fgcn_spatial_wgrad": (_I, [_P] * 4 + [_I] * 9 + [_P]),
    "fgcn_spatial_wgrad_tile": (_I, [_P] * 4 + [_I] * 8 + [_P]),
    "fgcn_spatial_wgrad_tile_slabs": (_I, [_I] * 5),
    "fgcn_spatial_wgrad_tile_available": (_I, [_I] * 3),
    "fgcn_emb_fwd_tile": (_I, [_P] * 5 + [_I] * 7 + [_P]),
    "fgcn_emb_fwd_tile_h": (_I, [_P] * 5 + [_I] * 7 + [_P]),
    "fgcn_emb_fwd_tile_segments": (_I, [_I] * 4),
    "fgcn_emb_fwd_tile_available": (_I, [_I] * 3),
    "fgcn_emb_dx_tile": (_I, [_P] * 5 + [_I] * 9 + [_P]),
    "fgcn_emb_dx_tile_h": (_I, [_P] * 5 + [_I] * 9 + [_P]),
    "fgcn_emb_dx_tile_workspace": (_LL, [_I, _I]),
    "fgcn_emb_wgrad_tile": (_I, [_P] * 5 + [_I] * 8 + [_P]),
    "fgcn_emb_wgrad_tile_h": (_I, [_P] * 5 + [_I] * 8 + [_P]),
    "fgcn_emb_wgrad_tile_slabs": (_I, [_I] * 5),
    "fgcn_emb_tile_available": (_I, [_I] * 3),
    "fgcn_joint_dagg": (_I, [_P] * 5 + [_I] * 11 + [_P] * 5),
    "fgcn_adj_softmax_fwd": (_I, [_P, _I, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fgcn_adj_softmax_bwd": (_I, [_P, _I, _F, _P, _P, _P, _I, _I, _I, _P]),
    "fgcn_bn_finalize": (_I, [_P, _I, _LL, _P, _P, _P, _P, _F, _F, _P, _I, _P]),
    "fgcn_bn_eval_coeffs": (_I, [_P, _P, _P, _P, _F, _P, _I, _P]),
    "fgcn_bn_act": (_I, [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _P]),
    "fgcn_tconv_halo_bn_relu": (_I, [_P] * 7 + [_I] * 10 + [_P]),
    "fgcn_spatial_fwd_tile_bn_relu": (_I, [_P] * 7 + [_I] + [_P] + [_I] * 8 + [_P]),
    "fgcn_bn_act_h": (_I, [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _P]),
    "fgcn_bn_act_bwd_apply_h": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _P]),
    "fgcn_tconv_halo_h": (_I, [_P, _P, _P, _P, _P] + [_I] * 18 + [_P, _P, _P] + [_P]),
    "fgcn_tconv_wgrad_h": (_I, [_P, _P, _P] + [_I] * 17 + [_P]),
    "fgcn_pw_wgrad_h": (_I, [_P, _P, _P] + [_I] * 11 + [_P]),
    "fgcn_spatial_bwd_tile_h": (_I, [_P, _P, _P, _P, _P, _P] + [_I] * 10 + [_P, _I, _P, _P, _P, _P]),
    "fgcn_spatial_wgrad_tile_h": (_I, [_P, _P, _P, _P] + [_I] * 8 + [_P]),
    # typed forms (half-precision activation storage, math mode bf16): `half_mask` before the stream
    "fgcn_bn_act_t": (_I, [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _P]),
    "fgcn_bn_act_pool_t": (_I, [_P] * 7 + [_I] * 5 + [_P]),
    "fgcn_bn_act_bwd_reduce_t": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _LL, _I, _I, _I, _I, _P]),
    "fgcn_bn_act_bwd_apply_t": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_tconv_halo_t": (_I, [_P, _P, _P, _P, _P] + [_I] * 18 + [_P]),
    "fgcn_spatial_fwd_tile_t": (_I, [_P] * 6 + [_I] * 9 + [_P]),
    "fgcn_emb_fwd_tile_t": (_I, [_P] * 5 + [_I] * 8 + [_P]),
    "fgcn_spatial_bwd_tile_t": (_I, [_P] * 6 + [_I] * 10 + [_P, _I, _P, _P, _P, _I, _P]),
    "fgcn_spatial_wgrad_tile_t": (_I, [_P] * 4 + [_I] * 9 + [_P]),
    "fgcn_emb_dx_tile_t": (_I, [_P] * 5 + [_I] * 9 + [_P, _I, _P]),
    "fgcn_emb_wgrad_tile_t": (_I, [_P] * 5 + [_I] * 9 + [_P]),
    "fgcn_rows_gemm_t": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, TMap, _I, _I, _P]),
    "fgcn_pw_gemm_t": (_I, [_P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_bn_act_pool_splits": (_I, [_I, _I]),
    "fgcn_bn_act_pool": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fgcn_bn_act_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _LL, _I, _I, _I, _P]),
    "fgcn_bn_act_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _P]),
    "fgcn_bn_act_bwd_reduce_g": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _LL, _I, _I, _I, _P]),
    "fgcn_bn_act_bwd_apply_g": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _P]),
    "fgcn_elem_tiles": (_I, [_LL]),
    "fgcn_bn_apply_ld": (_I, [_P, _P, _P, _LL, _I, _I, _P]),
    "fgcn_bn_bwd_reduce_ld": (_I, [_P, _I, _P, _P, _P, _I, _LL, _I, _P]),
    "fgcn_bn_bwd_apply_ld": (_I, [_P, _I, _P, _P, _P, _P, _LL, _I, _I, _P]),
    "fgcn_col_sum": (_I, [_P, _P, _LL, _I, _I, _P]),
    "fgcn_spatial_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_spatial_tiles": (_I, [_I, _I]),
    "fgcn_spatial_fwd_tile": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_spatial_fwd_tile_tiles": (_I, [_I, _I, _I]),
    "fgcn_spatial_fwd_tile_available": (_I, [_I, _I, _I]),
    "fgcn_spatial_bwd_tile": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "fgcn_spatial_bwd_tile_g": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    "fgcn_spatial_bwd_tile_segments": (_I, [_I, _I, _I]),
    "fgcn_spatial_bwd_tile_available": (_I, [_I, _I, _I]),
    "fgcn_transpose": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fgcn_row_softmax_fwd": (_I, [_P, _P, _P, _P, _LL, _I, _I, _I, _F, _P]),
    "fgcn_row_softmax_bwd": (_I, [_P, _P, _P, _LL, _I, _I, _F, _P]),
    "fgcn_tmaxpool3_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_tmaxpool3_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_unfold_windows": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_pw_gemm_available": (_I, []),
    "fgcn_pw_gemm_tiles": (_I, [_LL]),
    "fgcn_pw_gemm": (_I, [_P, _P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _P, _P]),
    "fgcn_data_bn_tiles": (_I, [_I, _I]),
    "fgcn_data_bn_stats": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fgcn_data_bn_apply": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_data_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_data_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fgcn_cross_entropy_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fgcn_cross_entropy_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fgcn_optim_step": (_I, [_P, _P, _P, _P, _LL, _I, _F, _F, _F, _F, _F, _F, _F, _F, _I, _LL, _P]),
}

_lib = None


class FgcnError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libfgcn.so once; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so); it must be the one already mapped when
        # libfgcn.so resolves its HIP symbols, otherwise two runtimes coexist and ours sees no device.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise FgcnError(f"{LIB_PATH} not found: build the HIP extension first (python -m fusion_gcn_amd.build). "
                            "There is no CPU / eager fallback for the AGCN block.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().fgcn_last_error()
        raise FgcnError(f"{what} failed ({rc}): {msg.decode(errors='replace') if msg else ''}")
