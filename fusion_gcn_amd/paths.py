"""Which kernel form a block takes for each stage: per-CONTEXT options, not process globals.

Until round 5 these were ~25 module attributes of block.py initialised from environment variables at import: per process, so two
models in one process could not differ in them and the `fgcn_ctx` guarantee ("two models in two modes do not see each other's
settings", include/fgcn.h) stopped at ops.py.  They now live in a ``PathOptions`` object carried by ``ops.Context``:

    with ops.context("bf16x3") as ctx:          # a fresh context = a copy of the thread's present settings
        ctx.paths.emb_tile = False                # this model's blocks run the unfused embedding chain
        out = model(x); out.sum().backward()      # (the backward runs in the forward's context, wherever autograd runs it)

``block.py`` reads ``ops.current_context().paths`` once per block call.  The process-wide default context takes its initial values
from ONE environment variable, ``FGCN_PATHS="name=value,name=value"`` (the same-call A/B scripts under tools/ set it per child
process); nothing else in the package reads the environment for a path decision.

Every default below is a measured choice; the measurement is cited next to the field (records under profiles/, narrative in
DESIGN.md / DESIGN_HISTORY.md).  Variants that lost their A/B twice were deleted in round 6 instead of carried as switches: the
four-wave spatial backward (tuning key 11 = 2) and the weight-gradient side stream.
"""
from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass, field
from typing import Dict, Tuple


@dataclass
class PathOptions:
    # -- 1x1 convolutions: the persistent split-bf16 row GEMM (ops.pw_gemm) from this contraction depth on, the exact-f32 row GEMM below.
    # MI355X, tools/kbench.py pw, B = 128 (profiles/r03_kbench_pw.log, r03_ab_pw_gemm.txt, r03_ab_pw_min_k_small.txt): at K = 64 / 96 the
    # f32 row GEMM is even or ahead in bf16x3 at 64 clips (58.0-58.4 vs 58.3-58.5 ms), pw_gemm wins from K = 128 (62.53 vs 62.82 ms) and,
    # with the f16x2 products or on few rows (the 8-clip shard: 9.65 -> 9.61 ms), from K = 64
    pw_min_k: int = 128
    pw_min_k_f16x2: int = 64
    pw_small_rows: int = 300_000
    # -- north-star kernel 2 in its stated form: BatchNorm + shortcut + ReLU of the graph convolution applied INSIDE the temporal conv
    # while it stages its image (ops.tconv_halo(fuse_in=...)); bit-identical to the two-pass form and measured SLOWER (63.10 / 63.16 ->
    # 64.68 / 64.96 ms at 64 clips, round 3; +2.0 ms re-measured in round 5: profiles/r03_ab_fused_input_stage.txt, r05_ab_fused_input_stage.txt)
    fuse_g: bool = False
    # -- fused spatial forward in its tile form from this many output channels on (profiles/r03_kbench_spatial_tile.log)
    # (per math mode: in `bf16` the matrix work is a sixth and the tile form's single pass over x wins from 64 outputs on --
    # profiles/r06_ab_bf16_paths.txt: 36.46 / 36.06 -> 35.83 / 35.72 ms)
    spatial_tile: bool = True
    spatial_tile_min_cout: Dict[str, int] = field(default_factory=lambda: {"f32": 128, "bf16": 64, "bf16x3": 128, "f16x2": 128})
    # -- the backward of the spatial stage in one launch (fgcn_spatial_bwd_tile.hip): 57.65 -> 56.0-56.2 ms (profiles/r04_ab_spatial_bwd_tile.txt)
    spatial_bwd_tile: bool = True
    spatial_bwd_tile_min_cin: int = 64
    spatial_bwd_tile_f16x2: bool = True          # ... also with the f16x2 products: 49.37 / 49.44 -> 48.75 / 48.78 ms
    fused_dagg: bool = True                      # (unfused route) dx mix + dA^ gram in one kernel
    # -- BatchNorm-backward sums of the graph convolution in the temporal data gradient's epilogue, up to this many channels
    # (bf16x3: neutral, round 4; `bf16`: the epilogue's dword loads of y and the sign image cost the now six times shorter conv more than
    # the stand-alone reduction pass takes -- profiles/r06_ab_bf16_paths.txt: 36.46 / 36.06 -> 35.01 / 35.05 ms with the pass)
    # (f16x2, re-measured with round 6's reduce kernel -- four rows of loads in flight: 46.46 / 46.54 -> 46.22 / 46.26 ms with the pass; bf16x3 still neutral,
    # 53.02 / 53.07 vs 53.06 / 53.01: profiles/r06_ab_bf16x3_paths.txt)
    bn_sums_in_dgrad: Dict[str, bool] = field(default_factory=lambda: {"f32": True, "bf16": False, "bf16x3": True, "f16x2": False})
    bn_sums_max_c: int = 4096
    # -- identity-shortcut gradients added by the kernel that forms the spatial term of dx: neutral in joint_dagg (off), 55.31 / 55.37 ->
    # 55.20 / 55.20 ms in the fused backward (profiles/r04_ab_gated_tile.txt)
    gated_shortcuts: bool = False
    gated_shortcuts_tile: bool = True
    # -- conv_d's weight gradient in tile form: 56.57 / 56.49 -> 55.71 / 55.53 ms (profiles/r04_ab_spatial_wgrad_tile.txt)
    spatial_wgrad_tile: bool = True
    spatial_wgrad_tile_f16x2: bool = True
    fused_agg_wgrad: bool = True                 # (older form) aggregation recomputed on chip, up to this many outputs per math mode
    fused_agg_wgrad_max_cout: Dict[str, int] = field(default_factory=lambda: {"f32": 64, "bf16": 128, "bf16x3": 128, "f16x2": 64})
    # -- attention embeddings: backward with the embedding gradient on chip, forward with the gram on chip; up to this many input
    # channels (profiles/r05_ab_emb_tile.txt: every block 53.79 / 53.90 ms, up to 128: 53.63 / 53.56, none: 54.09 / 54.22;
    # profiles/r05_ab_emb_fwd_tile.txt: off 54.98 / 54.93, up to 128: 54.73 / 54.87)
    # `bf16`: at 256 channels the tile kernels are no longer matrix-bound and the fused path wins there too
    # (profiles/r06_ab_bf16_paths.txt: 36.46 / 36.06 -> 36.10 / 35.89 ms)
    emb_tile: bool = True
    emb_tile_max_cin: Dict[str, int] = field(default_factory=lambda: {"f32": 128, "bf16": 256, "bf16x3": 128, "f16x2": 128})
    emb_fwd_tile: bool = True
    emb_fwd_tile_max_cin: Dict[str, int] = field(default_factory=lambda: {"f32": 128, "bf16": 256, "bf16x3": 128, "f16x2": 128})
    # ... and up to this many embedding channels per group (ic = out_channels / 4): at ic = 64 a workgroup holds theta_k | phi_k of ONE subset and x
    # is read three times -- the 128 -> 256 block's forward measures 0.587 ms in tile form against 0.431 ms for the 1x1 product + gram
    # (tools/kbench.py emb_fwd, bf16x3); in the step 53.13 / 52.97 -> 52.92 / 52.89 ms (bf16x3), 46.34 -> 46.15 / 46.25 (f16x2); in `bf16` the tile form
    # wins there by 1 ms (26.45 -> 27.5 ms without it): profiles/r06_ab_emb_fwd_tile_max_ic.txt
    emb_fwd_tile_max_ic: Dict[str, int] = field(default_factory=lambda: {"f32": 32, "bf16": 64, "bf16x3": 32, "f16x2": 32})
    # -- math mode bf16 (BASELINE config 5): half-precision STORAGE of the tensors that only bf16 MFMA staging reads -- G (the temporal conv's
    # input), dU (the gradient of its output) and dY (the gradient of the spatial stage's output) are written as bfloat16 by the BatchNorm
    # passes that produce them and copied by their consumers (include/fgcn.h, the `_h` entry points).  Bit-identical to f32 storage (the bf16 kernels round these tensors to bfloat16
    # when they stage them anyway); the halo conv, bound by its row traffic through L2 in this mode, moves half of it
    # (probe: 34.38 -> 32.9 ms from the conv's input alone, profiles/r06_ab_bf16_half_storage.txt).  Only the mode's own entry counts.
    half_storage: Dict[str, bool] = field(default_factory=lambda: {"f32": False, "bf16": True, "bf16x3": False, "f16x2": False})
    # -- math mode bf16, one step further: half-precision storage of EVERY activation-sized tensor of a training step -- the pre-BatchNorm
    # outputs Y / U, the block boundary x / O and its gradient, dG -- as the reference's autocast step keeps them (session/procedures/step.py:
    # 55-78); BatchNorm statistics, softmax, accumulators stay float32.  Unlike half_storage this changes VALUES (one more rounding per
    # stored tensor): SURVEY.md section 7's contract instead of bit-identity (tests/test_bf16_gpu.py; tools/probes/half_act_probe.py: logits
    # 2.3e-3 -> 2.2e-3, gradient cosine 0.9977 -> 0.9972 against the float32 path).  The typed `_t` entry points of include/fgcn.h.
    half_activations: Dict[str, bool] = field(default_factory=lambda: {"f32": False, "bf16": True, "bf16x3": False, "f16x2": False})
    # ... per producer (A/B switches inside half_activations): the temporal conv's output U, the spatial tile kernel's output Y
    half_conv_out: bool = True
    half_spatial_out: bool = True
    half_shortcuts: bool = True                  # the shortcut convolutions (down / residual) of the blocks that change width or stride: typed row GEMMs
    # -- inference (module in eval mode, autograd off): BatchNorm + shortcut + ReLU in the epilogues of the two north-star kernels
    # (fgcn_spatial_fwd_tile_bn_relu, fgcn_tconv_halo_bn_relu) -- a block is two kernels + the attention; split modes bf16x3 / bf16
    fused_inference: bool = True
    # -- the model's last block: the epilogue pass carries the global average pooling, and its backward reads the pooled gradient as one
    # row per clip (profiles/r05_ab_pool_epilogue_and_split_sums.txt, r05_ab_pool_backward_rows.txt: 53.74 / 53.74 -> 53.62 / 53.50 ms)
    pool_epilogue: bool = True
    pool_backward_rows: bool = True
    mix_vw_order: Tuple[int, ...] = (2, 1)       # channels per lane preference of the channel-group mix kernel

    def copy(self) -> "PathOptions":
        return dataclasses.replace(self, **{f.name: dict(getattr(self, f.name)) for f in dataclasses.fields(self)
                                            if isinstance(getattr(self, f.name), dict)})

    def get(self, name: str, mode: str):
        """The value of option ``name`` in math mode ``mode`` (per-mode options are dicts keyed by the mode's name)."""
        v = getattr(self, name)
        return v[mode] if isinstance(v, dict) else v

    def update_from(self, spec: str) -> "PathOptions":
        """``"name=value,name=value"`` (the FGCN_PATHS form).  Unknown names are an error: an A/B run that silently tests nothing is
        worse than one that stops.  Per-mode options: ``fused_agg_wgrad_max_cout=128`` sets every math mode's entry,
        ``emb_tile_max_cin.bf16=128`` one."""
        fields = {f.name: f for f in dataclasses.fields(self)}
        for item in filter(None, (s.strip() for s in spec.split(","))):
            name, _, val = item.partition("=")
            name, _, only_mode = name.strip().lower().partition(".")
            if name not in fields:
                raise ValueError(f"FGCN_PATHS: unknown path option {name!r} (known: {', '.join(sorted(fields))})")
            cur = getattr(self, name)
            if only_mode and not (isinstance(cur, dict) and only_mode in cur):
                raise ValueError(f"FGCN_PATHS: {name!r} has no entry for math mode {only_mode!r}")
            if isinstance(cur, bool):
                setattr(self, name, val.strip().lower() not in ("0", "false", "off", "no", ""))
            elif isinstance(cur, int):
                setattr(self, name, int(val))
            elif isinstance(cur, dict):
                sample = next(iter(cur.values()))
                new = (val.strip().lower() not in ("0", "false", "off", "no", "")) if isinstance(sample, bool) else int(val)
                setattr(self, name, {k: (new if (not only_mode or k == only_mode) else old) for k, old in cur.items()})
            elif isinstance(cur, tuple):
                setattr(self, name, tuple(int(v) for v in val.replace(":", " ").split()))
            else:  # pragma: no cover
                raise ValueError(f"FGCN_PATHS: cannot parse {name!r}")
        return self


def process_defaults() -> PathOptions:
    """The default context's options: the dataclass defaults, overridden by FGCN_PATHS (read once, at first use)."""
    return PathOptions().update_from(os.environ.get("FGCN_PATHS", ""))
