"""``mode: skeleton_imu_gcn_late_fusion`` -- the skeleton AGCN and the IMU graph model side by side, their pooled features
fused (concatenate / sum / ...) and classified by one ``fc`` (reference torch_src/models/mmargcn/late_fusion_models.py:45-75).
Both branches run on libfgcn kernels (agcn.Model; ImuGCN with either graph convolution); the fusion is a torch reduction over two (N, C) tensors, ``fc`` the row GEMM."""
import torch.nn as nn

from ...block import LinearFunction
from . import agcn
from . import imu_feature_models as imu_models
from .fusion import get_fusion, get_skeleton_imu_fusion_graph


class SkeletonImuGCN(nn.Module):
    """Two feature extractors (``agcn``: the skeleton clip, ``imu_gcn``: the inertial sequence as a graph), both without their
    classifier; ``fusion`` merges the two pooled vectors; ``fc`` classifies.  Same sub-module and state-dict names as the
    reference."""

    def __init__(self, data_shape, num_classes: int, graph, **kwargs):
        super().__init__()
        fusion_type = kwargs.get("fusion", "concatenate")
        if kwargs.pop("skeleton_imu_spatial_fusion", False):      # optionally also the IMU-joints skeleton graph
            graph = get_skeleton_imu_fusion_graph(graph, **kwargs)
        # the IMU branch always links signals across time steps and carries the extra top layer (reference :53-54)
        self.imu_gcn = imu_models.ImuGCN(data_shape, num_classes, **{**kwargs, "inter_signal_back_connections": True,
                                                                   "include_additional_top_layer": True, "without_fc": True})
        self.agcn = agcn.Model(data_shape["skeleton"], num_classes, graph, num_layers=kwargs.get("num_layers", 10),
                               without_fc=True, dropout=kwargs.get("dropout", 0.))
        self.fusion = get_fusion(fusion_type, concatenate_dim=-1)
        branches = 2 if fusion_type == "concatenate" else 1
        self.fc = nn.Linear(branches * self.agcn.out_channels, num_classes)

    def forward(self, x):
        features = self.fusion.combine(self.agcn(x["skeleton"]), self.imu_gcn(x["inertial"]))
        return LinearFunction.apply(features.contiguous(), self.fc.weight, self.fc.bias)
