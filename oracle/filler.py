"""Deterministic, library-independent tensor filler (TEST INFRASTRUCTURE).

A counter-based generator (splitmix64 over ``crc32(name) ^ salt`` and the element index) so the
golden-vector generator (which fills the *reference's* modules) and the tests (which fill the oracle and
the HIP-backed modules) produce bit-identical parameters and inputs without committing 3.45 M weights.
No reference code involved; values depend only on (name, salt, index).
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, lo: float = -1.0, hi: float = 1.0, salt: int = 0) -> np.ndarray:
    """float64 array of ``shape`` with entries uniform in [lo, hi), a pure function of (name, salt, index)."""
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64((zlib.crc32(name.encode()) << 16) ^ (salt & 0xFFFF) ^ ((salt >> 16) << 48))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed
    bits = _splitmix64(_splitmix64(ctr))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).reshape(shape)


def bellish(name: str, shape, scale: float = 1.0, salt: int = 0) -> np.ndarray:
    """Zero-mean, unit-variance-ish (sum of 4 uniforms) values times ``scale``."""
    acc = sum(uniform(f"{name}#{i}", shape, -1.0, 1.0, salt) for i in range(4))
    return acc * (scale * np.sqrt(3.0 / 4.0))


def fill_value_for(key: str, shape, salt: int = 0) -> np.ndarray:
    """Fill rule for one state-dict entry of an AGCN-family model, chosen by the key's suffix so that
    activations stay O(1) through 10 blocks and no parameter sits at its degenerate init value
    (gcn BN gamma = 1e-6, adj_b = 1e-6 in the reference — SURVEY.md Appendix C item 4)."""
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return uniform(key, shape, -0.2, 0.2, salt)
    if leaf == "running_var":
        return uniform(key, shape, 0.6, 1.4, salt)
    if leaf in ("adj_b", "PA"):
        return uniform(key, shape, -0.15, 0.15, salt)
    if leaf == "A_res":                             # MS-G3D's learnable residual adjacency (init +-1e-6 in the reference)
        return uniform(key, shape, -0.03, 0.03, salt)
    if leaf == "bias":
        return uniform(key, shape, -0.2, 0.2, salt)
    if leaf == "weight" and len(shape) == 1:       # BatchNorm gamma
        return uniform(key, shape, 0.6, 1.4, salt)
    if leaf == "weight":                            # conv (O, I, kt, 1) or linear (O, I)
        fan_in = int(np.prod(shape[1:]))
        return bellish(key, shape, scale=np.sqrt(1.0 / fan_in), salt=salt)
    raise KeyError(f"no fill rule for state-dict key {key!r}")


def fill_state_dict(state_dict, salt: int = 0, skip=("adj_a", "A"), prefix: str = "", rename=None) -> None:
    """In-place fill of every entry of a torch ``state_dict()`` except constant adjacency buffers.
    The fill value is a function of ``prefix + key`` (after ``rename``), so a sub-module filled with its
    full-model prefix gets exactly the values the full model would."""
    import torch

    with torch.no_grad():
        for key, tensor in state_dict.items():
            if key.rsplit(".", 1)[-1] in skip:
                continue
            name = prefix + key
            if rename is not None:
                name = rename(name)
            vals = fill_value_for(name, tuple(tensor.shape), salt)
            tensor.copy_(torch.from_numpy(np.ascontiguousarray(vals)).reshape(tensor.shape).to(tensor.dtype))


def skeleton_input(name: str, shape, salt: int = 0, empty_second_body: bool = False) -> np.ndarray:
    """Synthetic (N, M, T, V, C) skeleton clip batch in [-1, 1]; optionally zero the 2nd body of odd clips."""
    x = bellish(name, shape, scale=0.5, salt=salt)
    if empty_second_body and shape[1] > 1:
        x[1::2, 1:] = 0.0
    return x
