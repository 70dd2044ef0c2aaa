"""Skeleton + IMU fusion graph (host side, build time).

Mirrors ``get_skeleton_imu_fusion_graph`` of the reference (torch_src/models/mmargcn/fusion.py:65-89):
IMU modalities become ``num_imu_joints`` extra vertices V..V+n-1 attached either to the skeleton's
centre joint ("append_center") or to right wrist + right hip ("append_right"), optionally pairwise
inter-connected.  The ``Fusion`` combiners of that file (:9-62; used by the late-fusion models on the two pooled feature
vectors) are one-line torch reductions and are mirrored as they are.
"""
import functools

import torch

from ...util.graph import Graph


def _weighted(stack: torch.Tensor, weights) -> torch.Tensor:
    return (stack * weights).sum(-1)


# fusion type -> reduction over the tuple of equally shaped feature tensors (stacked on a new last axis where needed)
_COMBINERS = {
    "sum": lambda ts, o: functools.reduce(torch.add, ts),
    "product": lambda ts, o: functools.reduce(torch.mul, ts),
    "average": lambda ts, o: torch.stack(ts, dim=-1).mean(-1),
    "weighted_average": lambda ts, o: _weighted(torch.stack(ts, dim=-1), o["weights"]),
    "concatenate": lambda ts, o: torch.cat(ts, dim=o["concatenate_dim"]),
}
_OPTIONS = {"weighted_average": ("weights",), "concatenate": ("concatenate_dim",)}


class Fusion:
    """``combine(*tensors)`` of the reference's Fusion classes, as one object: the fusion type picks the reduction."""

    def __init__(self, fusion_type: str, **options):
        self.fusion_type, self.options = fusion_type, options

    def combine(self, *tensors: torch.Tensor) -> torch.Tensor:
        return _COMBINERS[self.fusion_type](tensors, self.options)


def get_fusion(fusion_type: str, **kwargs) -> Fusion:
    """Same call as the reference's factory: options a fusion type does not take are ignored (e.g. ``concatenate_dim`` for "sum")."""
    if fusion_type not in _COMBINERS:
        raise ValueError("Unsupported fusion: " + fusion_type)
    wanted = _OPTIONS.get(fusion_type, ())
    missing = [k for k in wanted if k not in kwargs]
    if missing:
        raise TypeError(f"fusion {fusion_type!r} needs {missing}")
    return Fusion(fusion_type, **{k: kwargs[k] for k in wanted})


def get_skeleton_imu_fusion_graph(skeleton_graph: Graph, imu_enhanced_mode: str, num_imu_joints: int, **kwargs) -> Graph:
    first = skeleton_graph.num_vertices
    imu = range(first, first + num_imu_joints)
    if imu_enhanced_mode == "append_center":
        anchor = kwargs.get("center_joint", skeleton_graph.center_joint)
        new_edges = [(j, anchor) for j in imu]
    elif imu_enhanced_mode == "append_right":
        wrist, hip = kwargs["right_wrist_joint"], kwargs["right_hip_joint"]
        new_edges = [(j, a) for j in imu for a in (wrist, hip)]
    else:
        raise ValueError("Unsupported imu_enhanced_mode: " + imu_enhanced_mode)
    if kwargs.get("interconnect_imu_joints", False):
        new_edges += [(a, b) for a in imu for b in imu if a < b]
    return skeleton_graph.with_new_edges(new_edges)
