#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generate tests/golden/imu_gcn.npz by IMPORTING THE REFERENCE (build container only): IMU graph
adjacencies of build_imu_graph_adjacency and ImuGCN (gc_model "stgcn") logits / loss / gradients on filler-generated
parameters and inputs.  Same import recipe as oracle/gen_golden.py.  Run: python oracle/gen_golden_imu.py"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np  # noqa: E402

np.int = int
np.float = float
_tv = types.ModuleType("torchvision")
_tv.models = types.ModuleType("torchvision.models")
sys.modules.update({"torchvision": _tv, "torchvision.models": _tv.models, "cv2": types.ModuleType("cv2")})
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FGCN_REFERENCE", "/root/reference")
sys.path[:0] = [REF, os.path.join(REF, "torch_src")]
sys.path.append(REPO)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import models.mmargcn.imu_feature_models as ref_imu  # noqa: E402  (reference)
import models.mmargcn.mmargcn as ref_mm  # noqa: E402  (reference)
from util.graph import Graph  # noqa: E402  (reference)
from datasets.utd_mhad import constants as utd  # noqa: E402  (reference)

from oracle import filler  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
CASES = {   # tag: (data_shape, classes, batch, model kwargs)
    "value48": ((8, 6), 7, 3, dict(gc_model="stgcn", graph_node_format="node_per_value", num_layers=5, inner_feature_dim=16)),
    "sensor16": ((8, 6), 5, 4, dict(gc_model="stgcn", graph_node_format="node_per_sensor", num_signals=2, num_layers=4,
                                    inner_feature_dim=8, adjacency_normalization="row", num_temporal_back_connections=2,
                                    inter_signal_back_connections=True)),
    "agcn_sensor16": ((8, 6), 5, 3, dict(gc_model="agcn", graph_node_format="node_per_sensor", num_signals=2, num_layers=4,
                                         inner_feature_dim=16)),
    "agcn_value48": ((8, 6), 7, 2, dict(gc_model="agcn", graph_node_format="node_per_value", num_layers=3, inner_feature_dim=16,
                                        inter_signal_back_connections=True)),
}


LATE_KW = dict(gc_model="stgcn", graph_node_format="node_per_sensor", num_signals=2, num_layers=4, inner_feature_dim=64)


def main():
    store = {}
    for tag, (shape, sig, kw) in {"row_t1": ((5, 3), 0, dict(normalization="row")),
                                  "column_t2_inter": ((5, 3), 0, dict(normalization="column", temporal_back_connections=2,
                                                                       inter_signal_back_connections=True)),
                                  "symmetric_sensor": ((6, 4), 2, dict(normalization="symmetric"))}.items():
        store[f"adj.{tag}"] = ref_imu.build_imu_graph_adjacency(shape, sig, "stgcn", False, **kw).double().numpy()
    for tag, (shape, classes, batch, kw) in CASES.items():
        model = ref_imu.ImuGCN({"inertial": shape}, classes, **kw).double()
        filler.fill_state_dict(model.state_dict(), skip=("adj", "adj_a"))
        x = torch.from_numpy(filler.bellish(f"x.{tag}", (batch, *shape), scale=0.5)).double()
        labels = torch.from_numpy(filler.uniform(f"y.{tag}", (batch,), 0, classes).astype(np.int64))
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        model.eval()
        store[f"{tag}.eval.logits"] = model(x).detach().numpy()
        model.train()
        logits = model(x)
        loss = F.cross_entropy(logits, labels)
        loss.backward()
        store[f"{tag}.labels"] = labels.numpy()
        store[f"{tag}.train.logits"] = logits.detach().numpy()
        store[f"{tag}.train.loss"] = loss.detach().numpy()
        for name, p in model.named_parameters():
            store[f"{tag}.grad.{name}"] = p.grad.numpy()
        for k, v in model.state_dict().items():
            if k.endswith(("running_mean", "running_var")):
                store[f"{tag}.after.{k}"] = v.numpy().copy()
        store[f"{tag}.keys"] = np.array(sorted(sd0))
        store[f"{tag}.adj"] = sd0["gcn.gc1.adj" if "gcn.gc1.adj" in sd0 else "gcn.gc1.adj_a"].numpy()
    # ---- mode skeleton_imu_gcn_late_fusion (late_fusion_models.py:45-75), IMU branch stgcn / agcn ------------------------------
    for tag, late_kw in (("late", LATE_KW), ("late_agcn", dict(LATE_KW, gc_model="agcn"))):
        classes, batch = 27, 3
        shapes = {"skeleton": (1, 16, 20, 3), "inertial": (8, 6)}
        model = ref_mm.Model(shapes, classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint),
                             mode="skeleton_imu_gcn_late_fusion", **late_kw).double()
        filler.fill_state_dict(model.state_dict(), skip=("adj", "adj_a", "A"), rename=lambda k: k.replace("_model.", ""))
        x = {"skeleton": torch.from_numpy(filler.skeleton_input("x.late.skeleton", (batch, *shapes["skeleton"]))).double(),
             "inertial": torch.from_numpy(filler.bellish("x.late.inertial", (batch, *shapes["inertial"]), scale=0.5)).double()}
        labels = torch.from_numpy(filler.uniform("y.late", (batch,), 0, classes).astype(np.int64))
        store[f"{tag}.keys"] = np.array(sorted(k.replace("_model.", "") for k in model.state_dict()))
        model.eval()
        store[f"{tag}.eval.logits"] = model(x).detach().numpy()
        model.train()
        logits = model(x)
        loss = F.cross_entropy(logits, labels)
        loss.backward()
        store[f"{tag}.labels"] = labels.numpy()
        store[f"{tag}.train.logits"] = logits.detach().numpy()
        store[f"{tag}.train.loss"] = loss.detach().numpy()
        for name, p in model.named_parameters():
            store[f"{tag}.gl2.{name.replace('_model.', '')}"] = p.grad.norm().numpy()
    # ---- mode skeleton_imu_channel_fusion (early_fusion_models.py:25-45): IMU signals as extra channels of every joint ----------
    shapes, classes, batch = {"skeleton": (1, 16, 20, 3), "inertial": (16, 6)}, 27, 2
    model = ref_mm.Model(shapes, classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint),
                         mode="skeleton_imu_channel_fusion", num_layers=4).double()
    filler.fill_state_dict(model.state_dict(), skip=("adj_a", "A"), rename=lambda k: k.replace("_model.agcn.", ""))
    x = {"skeleton": torch.from_numpy(filler.skeleton_input("x.chan.skeleton", (batch, *shapes["skeleton"]))).double(),
         "inertial": torch.from_numpy(filler.bellish("x.chan.inertial", (batch, *shapes["inertial"]), scale=0.5)).double()}
    model.eval()
    store["chan.eval.logits"] = model(x).detach().numpy()
    model.train()
    store["chan.train.logits"] = model(x).detach().numpy()
    store["chan.keys"] = np.array(sorted(k.replace("_model.", "") for k in model.state_dict()))
    store["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, "imu_gcn.npz"), **store)
    print("wrote", os.path.join(OUT, "imu_gcn.npz"), os.path.getsize(os.path.join(OUT, "imu_gcn.npz")), "bytes")


if __name__ == "__main__":
    main()
