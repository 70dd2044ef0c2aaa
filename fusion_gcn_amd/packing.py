"""Packed / split weight forms as data, refreshed for a whole model in one launch (fgcn_pack_run, include/fgcn.h).

The kernels stream weights in layouts of their own (k-interleaved float4, three-way bfloat16 splits in fragment or accumulator
order, transposed / concatenated 1x1 matrices ...).  A ``Form`` states WHAT a layout contains -- the logical matrix
``W[tap][k][n]`` as a sum of strided windows (``Seg``) of parameter tensors -- instead of building it with torch ops; the
device table of all forms of all blocks is walked by one kernel launch per optimizer step (``PackPlan.run``), where the
previous host code issued ~170 tiny launches (cat / permute / contiguous / pack_split3) on the step's critical path.

Reference: the matrices are the reference's Conv2d weights (torch_src/models/mmargcn/agcn.py:41-42,71-73,77), which ATen
consumes in place; nothing here changes their values.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import PACK_MAX_SEG, PACK_MODES, PackItem, PackSeg


@dataclass
class Seg:
    """Window [t0, t0+tlen) x [k0, k0+klen) x [n0, n0+nlen) of the logical matrix, read from ``src`` (a contiguous float32
    parameter) at element ((tap-t0)*tap_step + tap0)*st_tap + (k-k0)*st_k + (n-n0)*st_n."""
    src: torch.Tensor
    st_k: int
    st_n: int
    klen: int
    nlen: int
    k0: int = 0
    n0: int = 0
    st_tap: int = 0
    t0: int = 0
    tlen: int = 1
    tap0: int = 0
    tap_step: int = 1


@dataclass
class Form:
    mode: str                   # "plain" | "k4" | "split3" | "split3_acc"
    taps: int
    K: int
    N: int
    segs: List[Seg]
    shape: Optional[Tuple[int, ...]] = None      # view of the plain / k4 result handed to the consumer (default: by mode)
    dst: Optional[torch.Tensor] = field(default=None, repr=False)
    packed: Optional[tuple] = field(default=None, repr=False)    # stamp() of the sources when dst was last packed (stamping plans)

    def stamp(self) -> tuple:
        """Where this form's sources live and which version they hold now (parameters: the Seg keeps the Parameter object, so a
        moved ``.data`` and an in-place update both show)."""
        return tuple((s.src.data_ptr(), s.src._version) for s in self.segs)

    def alloc(self, device) -> torch.Tensor:
        lib = _lib.load()
        kg = lib.fgcn_pack_kgroups(PACK_MODES[self.mode], self.K)
        if self.mode in ("split2h", "split2h_acc"):
            from .ops import ScaledWeights
            self.dst = ScaledWeights(self.taps, self.K, self.N, device, acc_order=self.mode == "split2h_acc")
        elif self.mode in ("split3", "split3_acc"):
            self.dst = torch.empty((3, self.taps, kg, self.N, 8), device=device, dtype=torch.bfloat16)
        elif self.mode == "k4":
            self.dst = torch.empty(self.shape or (self.taps, kg, self.N, 4), device=device, dtype=torch.float32)
        else:
            self.dst = torch.empty(self.shape or (self.taps, self.K, self.N), device=device, dtype=torch.float32)
        return self.dst

    def item(self) -> PackItem:
        if len(self.segs) > PACK_MAX_SEG:
            raise _lib.FgcnError(f"a packed form takes at most {PACK_MAX_SEG} segments")
        lib = _lib.load()
        it = PackItem()
        it.dst, it.mode, it.taps, it.K, it.N = self.dst.data_ptr(), PACK_MODES[self.mode], self.taps, self.K, self.N
        it.kgroups, it.nseg = lib.fgcn_pack_kgroups(it.mode, self.K), len(self.segs)
        for i, s in enumerate(self.segs):
            if not (s.src.is_contiguous() and s.src.dtype == torch.float32 and s.src.device == self.dst.device):
                raise _lib.FgcnError("packed forms read contiguous float32 parameters on the form's device")
            last = (s.tlen - 1) * s.tap_step * s.st_tap + s.tap0 * s.st_tap + (s.klen - 1) * s.st_k + (s.nlen - 1) * s.st_n
            if s.klen <= 0 or s.nlen <= 0 or s.tlen <= 0 or last >= s.src.numel() or s.k0 + s.klen > self.K or s.n0 + s.nlen > self.N \
                    or s.t0 + s.tlen > self.taps:
                raise _lib.FgcnError(f"segment {i} reaches outside its parameter or the form ({s.klen}x{s.nlen} at {s.k0},{s.n0})")
            it.seg[i] = PackSeg(s.src.data_ptr(), s.st_tap, s.st_k, s.st_n, s.t0, s.tlen, s.k0, s.klen, s.n0, s.nlen, s.tap0,
                                s.tap_step)
        return it

    def units(self) -> int:
        return int(_lib.load().fgcn_pack_units(PACK_MODES[self.mode], self.taps, self.K, self.N))


class PackPlan:
    """The device-side work list of a set of forms; ``run()`` rebuilds every one of them in ONE launch on the current stream."""

    def __init__(self, forms: Sequence[Form], stamp: bool = False):
        self.forms = [f for f in forms if f.dst is not None]
        self.n_wg = 0
        self.stamp = stamp            # record Form.packed on every run (fops.ParamForms checks it; the AGCN blocks track versions themselves)
        if not self.forms:
            return
        dev = self.forms[0].dst.device
        items = (PackItem * len(self.forms))(*[f.item() for f in self.forms])
        blockmap: List[int] = []
        for i, f in enumerate(self.forms):
            for b in range((f.units() + 255) // 256):
                blockmap += (i, b)
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        self.items_dev = raw.to(dev)
        self.map_dev = torch.tensor(blockmap, dtype=torch.int32).to(dev)
        self.n_wg = len(blockmap) // 2
        self.scaled = any(f.mode in ("split2h", "split2h_acc") for f in self.forms)    # FGCN_PACK_SPLIT2H items: per-form maxima first
        self.srcs = [s.src for f in self.forms for s in f.segs]        # the table holds raw pointers: keep the tensors alive

    def run(self) -> None:
        if self.n_wg:
            stream = torch.cuda.current_stream(self.items_dev.device).cuda_stream
            if self.scaled:
                _lib.check(_lib.load().fgcn_pack_run_scaled(self.items_dev.data_ptr(), self.map_dev.data_ptr(), self.n_wg,
                                                            len(self.forms), stream), "fgcn_pack_run_scaled")
            else:
                _lib.check(_lib.load().fgcn_pack_run(self.items_dev.data_ptr(), self.map_dev.data_ptr(), self.n_wg, stream),
                           "fgcn_pack_run")
            if self.stamp:
                for f in self.forms:
                    f.packed = f.stamp()


class PackedWeights:
    """The forms of one block: a form is materialised on first use (its own small launch) and from then on refreshed with all
    the others.  ``key in W`` says whether the form exists for this block and math mode; ``W[key]`` / ``W.get(key)`` hand the
    consumer the buffer; the buffers keep their addresses for the life of the object (HIP-graph capture relies on it)."""

    def __init__(self, specs: Dict[str, Form], device):
        self.specs, self.device = specs, device
        self.live: Dict[str, Form] = {}
        self._plan: Optional[PackPlan] = None
        self.fresh = False            # the live forms hold the current parameter values

    def __contains__(self, key) -> bool:
        return key in self.specs

    def __getitem__(self, key) -> torch.Tensor:
        f = self.live.get(key)
        if f is None:
            f = self.specs[key]
            f.alloc(self.device)
            PackPlan([f]).run()
            self.live[key] = f
            self._plan = None
        return f.dst

    def get(self, key, default=None):
        return self[key] if key in self.specs else default

    def live_forms(self) -> List[Form]:
        return list(self.live.values())

    def refresh(self) -> None:
        """Re-pack this block's live forms (a standalone block; a Model refreshes all its blocks with one plan instead)."""
        if self._plan is None or len(self._plan.forms) != len(self.live):
            self._plan = PackPlan(self.live_forms())
        self._plan.run()
        self.fresh = True
