"""TEST INFRASTRUCTURE -- ReLU-decision accounting between the HIP model and the float64 oracle.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this.  Why it exists: the end-to-end gradient of the
10-block model is a discontinuous function of the 20 ReLU sign patterns.  A pre-activation that sits within float32 rounding of
zero can come out positive in one correct float32 implementation and non-positive in another (or in the float64 oracle); each
such flip zeroes / un-zeroes one upstream gradient element, and the flat gradient then differs at the 1e-3 level although
every kernel is accurate to 1e-6.  So gradient parity is stated in three parts:

  (i)   the ReLU decisions of the HIP forward are compared with the oracle's, block by block (``count_flips``);
  (ii)  the HIP backward is re-run with the ORACLE's decisions written into the one-bit sign images it gates on
        (``inject``): every remaining difference is then arithmetic, and the flat gradient must agree to <= 1e-4;
  (iii) the un-injected error must be explained by the flips (``explained_by_flips``: the part of the error that lives
        outside the injected run's error is attributed to them and bounded).

The sign image layout is fgcn_bn_act's (include/fgcn.h): bit e % 8 of byte e / 8 = [out[e] > 0] over the flat channels-last
(B, T, V, C) tensor.  The oracle works in the reference's (B, C, T, V) layout (torch_src/models/mmargcn/agcn.py:186-188).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

RELU_NAMES = ("g", "o")      # G = relu(BN(y) + down(x)) (agcn.py:113-115), O = relu(tcn(G) + residual(x)) (agcn.py:135-136)


def pack_sign_image(t_nchw: torch.Tensor) -> np.ndarray:
    """[t > 0] of an oracle (B, C, T, V) tensor as the kernels' bit image (uint8, numel / 8) of the (B, T, V, C) layout."""
    bits = (t_nchw.permute(0, 2, 3, 1) > 0).contiguous().numpy().reshape(-1)
    return np.packbits(bits, bitorder="little")


def oracle_sign_images(capture: Dict[str, torch.Tensor], num_blocks: int) -> List[Dict[str, np.ndarray]]:
    """capture = the dict agcn_oracle.model_forward(..., capture=) filled -> per block {"g": bits, "o": bits}."""
    return [{n: pack_sign_image(capture[f"l{i}.{n}"]) for n in RELU_NAMES} for i in range(num_blocks)]


class BlockTaps:
    """Forward hooks on the HIP model's blocks that keep each block's autograd node (= the STBlockFunction ctx, whose
    ``S`` dict holds the saved sign images ``g_sign`` / ``o_sign`` the backward kernels gate on)."""

    def __init__(self, model):
        from fusion_gcn_amd.models.mmargcn.agcn import SpatialTemporalConv
        self.nodes: List[object] = []
        self.handles = []
        for m in model.modules():
            if isinstance(m, SpatialTemporalConv):
                self.handles.append(m.register_forward_hook(lambda _m, _i, out: self.nodes.append(out.grad_fn)))

    def reset(self) -> None:
        self.nodes.clear()

    def close(self) -> None:
        for h in self.handles:
            h.remove()

    def sign_images(self) -> List[Dict[str, np.ndarray]]:
        out = []
        for node in self.nodes:
            S = node.S
            if S["g_sign"] is None or S["o_sign"] is None:
                raise RuntimeError("block has no sign image (element count not a multiple of 8)")
            out.append({"g": S["g_sign"].cpu().numpy().copy(), "o": S["o_sign"].cpu().numpy().copy()})
        return out

    def inject(self, images: List[Dict[str, np.ndarray]]) -> None:
        """Overwrite the saved sign images with the oracle's: the backward that follows gates on the oracle's ReLU decisions."""
        assert len(images) == len(self.nodes)
        for node, im in zip(self.nodes, images):
            for n in RELU_NAMES:
                dst = node.S[f"{n}_sign"]
                src = torch.from_numpy(im[n])
                assert dst.numel() == src.numel(), (dst.numel(), src.numel())
                dst.copy_(src.to(dst.device))


_POP = np.array([bin(i).count("1") for i in range(256)], dtype=np.int64)


def count_flips(hip: List[Dict[str, np.ndarray]], ora: List[Dict[str, np.ndarray]]):
    """-> (per-block [(flips_g, flips_o)], total flips, total ReLU decisions)."""
    per, total, n = [], 0, 0
    for h, o in zip(hip, ora):
        row = []
        for name in RELU_NAMES:
            f = int(_POP[np.bitwise_xor(h[name], o[name])].sum())
            row.append(f)
            total += f
            n += 8 * h[name].size
        per.append(tuple(row))
    return per, total, n


def flat_grads(model) -> torch.Tensor:
    return torch.cat([p.grad.detach().double().flatten().cpu() for _, p in model.named_parameters()])


def oracle_side(x, labels, sd, names):
    """The oracle's half of the report, computed once: logits, loss, the flat gradient in ``names`` order (the HIP model's
    named_parameters order) and the per-block ReLU sign images.  ``x`` / ``sd`` float64 for the strict comparison; bench.py's
    cpu_baseline leg hands in the float32 run it times anyway."""
    from oracle import agcn_oracle as O
    cap: Dict[str, torch.Tensor] = {}
    lo, los, grads_o, _ = O.loss_and_grads(x, labels, sd, capture=cap)
    flat_o = torch.cat([grads_o[n].double().flatten() for n in names])
    nblocks = sum(1 for k in cap if k.endswith(".g"))
    images = oracle_sign_images(cap, nblocks)
    return dict(logits=lo.double(), loss=float(los), flat=flat_o, images=images)


def gradient_parity_report(model, x_dev, labels_dev, x64=None, labels=None, sd64=None, oracle=None, keep_grads: bool = False):
    """Runs (i)-(iii) for one case.  Returns a dict: flips (per block, total, decisions), err_plain (HIP backward as is),
    err_injected (HIP backward gated on the oracle's ReLU decisions), both flat-gradient rel-L2 against the oracle,
    and logits / loss errors of the forward.  ``oracle``: a precomputed ``oracle_side`` result (else computed from x64, labels, sd64).
    ``keep_grads``: leave the gradients of the LAST (injected) backward in ``p.grad``."""
    from fusion_gcn_amd.loss import cross_entropy      # the product's loss (libfgcn), checked against the oracle's F.cross_entropy

    if oracle is None:
        oracle = oracle_side(x64, labels, sd64, [n for n, _ in model.named_parameters()])
    lo, los, flat_o, ora = oracle["logits"], oracle["loss"], oracle["flat"], oracle["images"]
    taps = BlockTaps(model)
    out = {}
    try:
        for tag in ("plain", "injected"):
            taps.reset()
            model.zero_grad(set_to_none=True)
            logits = model(x_dev)
            loss = cross_entropy(logits, labels_dev)
            if tag == "plain":
                hip = taps.sign_images()
                per, total, n = count_flips(hip, ora)
                out.update(flips_per_block=per, flips=total, decisions=n,
                           logits_err=float((logits.detach().double().cpu() - lo).norm() / lo.norm()),
                           loss_err=abs(float(loss.detach()) - float(los)))
            else:
                taps.inject(ora)
            loss.backward()
            torch.cuda.synchronize()
            out[f"err_{tag}"] = float((flat_grads(model) - flat_o).norm() / flat_o.norm())
    finally:
        taps.close()
        if not keep_grads:
            model.zero_grad(set_to_none=True)
    return out
