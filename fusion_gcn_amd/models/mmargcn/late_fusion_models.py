"""``mode: skeleton_imu_gcn_late_fusion`` -- the skeleton AGCN and the IMU graph model side by side, their pooled features
fused (concatenate / sum / ...) and classified by one ``fc`` (reference torch_src/models/mmargcn/late_fusion_models.py:45-75).
Both branches run on libfgcn kernels (agcn.Model; ImuGCN with either graph convolution); the fusion is a torch reduction over two (N, C) tensors, ``fc`` the row GEMM."""
import torch.nn as nn

from ...block import LinearFunction
from . import agcn
from . import imu_feature_models as imu_models
from .fusion import get_fusion, get_skeleton_imu_fusion_graph


class SkeletonImuGCN(nn.Module):
    def __init__(self, data_shape, num_classes: int, graph, **kwargs):
        super().__init__()
        num_layers = kwargs.get("num_layers", 10)
        dropout = kwargs.get("dropout", 0.)
        fusion_type = kwargs.get("fusion", "concatenate")
        if kwargs.pop("skeleton_imu_spatial_fusion", False):
            graph = get_skeleton_imu_fusion_graph(graph, **kwargs)
        self.imu_gcn = imu_models.ImuGCN(data_shape, num_classes, inter_signal_back_connections=True,
                                         include_additional_top_layer=True, without_fc=True, **kwargs)
        self.agcn = agcn.Model(data_shape["skeleton"], num_classes, graph, num_layers=num_layers, without_fc=True,
                               dropout=dropout)
        self.fusion = get_fusion(fusion_type, concatenate_dim=-1)
        out_dim = self.agcn.out_channels * 2 if fusion_type == "concatenate" else self.agcn.out_channels
        self.fc = nn.Linear(out_dim, num_classes)

    def forward(self, x):
        skeleton_data = self.agcn(x["skeleton"])
        inertial_data = self.imu_gcn(x["inertial"])
        fused_data = self.fusion.combine(skeleton_data, inertial_data)
        return LinearFunction.apply(fused_data.contiguous(), self.fc.weight, self.fc.bias)
