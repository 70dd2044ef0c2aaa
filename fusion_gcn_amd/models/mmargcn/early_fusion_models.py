"""Early skeleton + IMU fusion in front of one AGCN (reference torch_src/models/mmargcn/early_fusion_models.py:9-45).

``mode: skeleton_imu_spatial_fusion`` -- ``SkeletonImuSpatialFusionModel`` (:9-22): every IMU modality is an extra joint of the
skeleton graph (the preprocessing already appended their samples as joints V..V+n-1), so only the GRAPH changes.
``mode: skeleton_imu_channel_fusion`` -- ``SkeletonImuChannelFusionModel`` (:25-45): the IMU signals of a frame are broadcast to
every joint of every body as extra input CHANNELS, so only the input changes.
The RGB early-fusion variants of that file wrap image encoders and are out of scope.

Both are one AGCN (sub-module ``agcn``: the state-dict prefix the reference's checkpoints carry) behind a small adapter; what
differs is stated as two hooks instead of two constructors.
"""
import torch
import torch.nn as nn

from . import agcn
from .fusion import get_skeleton_imu_fusion_graph

# model_args the reference forwards to agcn.Model (defaults are agcn.Model's own), plus this build's block switches
_FORWARDED = ("num_layers", "without_fc")
_BUILD_SWITCHES = ("static_adjacency", "fused_spatial")


class _AgcnBehindAdapter(nn.Module):
    forwarded = _FORWARDED

    def __init__(self, data_shape, num_classes: int, graph, **kwargs):
        super().__init__()
        shape, graph = self.network_shape_and_graph(data_shape, graph, kwargs)
        passed = {k: kwargs[k] for k in self.forwarded if k in kwargs}
        self.agcn = agcn.Model(tuple(shape), num_classes, graph, **passed)

    def network_shape_and_graph(self, data_shape, graph, kwargs):
        raise NotImplementedError

    def network_input(self, x):
        return x

    def forward(self, x):
        return self.agcn(self.network_input(x))


class SkeletonImuSpatialFusionModel(_AgcnBehindAdapter):
    forwarded = _FORWARDED + _BUILD_SWITCHES

    def network_shape_and_graph(self, data_shape, graph, kwargs):
        return data_shape["skeleton"], get_skeleton_imu_fusion_graph(graph, **kwargs)


class SkeletonImuChannelFusionModel(_AgcnBehindAdapter):
    def network_shape_and_graph(self, data_shape, graph, kwargs):
        *lead, channels = data_shape["skeleton"]
        return (*lead, channels + data_shape["inertial"][-1]), graph

    def network_input(self, x):
        skeleton, imu = x["skeleton"], x["inertial"]                  # (N, M, T, V, C) and (N, T, S): frames line up
        n, bodies, frames, joints, _ = skeleton.shape
        imu_on_joints = imu[:, None, :, None, :].expand(n, bodies, frames, joints, imu.shape[-1])
        return torch.cat((skeleton, imu_on_joints), dim=-1)
