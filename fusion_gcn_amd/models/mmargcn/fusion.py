"""Skeleton + IMU fusion graph (host side, build time).

Mirrors ``get_skeleton_imu_fusion_graph`` of the reference (torch_src/models/mmargcn/fusion.py:65-89):
IMU modalities become ``num_imu_joints`` extra vertices V..V+n-1 attached either to the skeleton's
centre joint ("append_center") or to right wrist + right hip ("append_right"), optionally pairwise
inter-connected.  The ``Fusion`` combiners of that file (:9-62; used by the late-fusion models on the two pooled feature
vectors) are one-line torch reductions and are mirrored as they are.
"""
import functools
import inspect

import torch

from ...util.graph import Graph


class Fusion:
    def combine(self, *tensors: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class SumFusion(Fusion):
    def combine(self, *tensors):
        return functools.reduce(torch.add, tensors)


class ProductFusion(Fusion):
    def combine(self, *tensors):
        return functools.reduce(torch.mul, tensors)


class AverageFusion(Fusion):
    def combine(self, *tensors):
        return torch.mean(torch.stack(tensors, dim=-1), dim=-1)


class WeightedAverageFusion(Fusion):
    def __init__(self, weights: torch.Tensor):
        self.weights = weights

    def combine(self, *tensors):
        return torch.sum(torch.stack(tensors, dim=-1) * self.weights, dim=-1)


class ConcatenateFusion(Fusion):
    def __init__(self, concatenate_dim: int):
        self._dim = concatenate_dim

    def combine(self, *tensors):
        return torch.cat(tensors, dim=self._dim)


def get_fusion(fusion_type: str, **kwargs) -> Fusion:
    fusion_types = {"sum": SumFusion, "product": ProductFusion, "concatenate": ConcatenateFusion,
                    "average": AverageFusion, "weighted_average": WeightedAverageFusion}
    if fusion_type not in fusion_types:
        raise ValueError("Unsupported fusion: " + fusion_type)
    args = inspect.getfullargspec(fusion_types[fusion_type].__init__).args
    return fusion_types[fusion_type](**{k: v for k, v in kwargs.items() if k in args})


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
