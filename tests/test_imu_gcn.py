"""SURVEY.md section 8 row f1 (static-adjacency half): the IMU graph model ``mode: imu_gcn`` with ``gc_model: stgcn``.

Chain of evidence: tests/golden/imu_gcn.npz is written by oracle/gen_golden_imu.py from the imported reference
(ImuGCN, build_imu_graph_adjacency); the CPU tests pin the oracle restatement and the product's graph builder / state-dict
surface to it; the GPU tests compare the HIP-backed model with the float64 oracle (forward 2e-5, gradients 5e-4 rel-L2,
exactly-zero gradients of the conv biases in front of a train-mode BatchNorm)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import filler
from oracle import imu_gcn_oracle as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "imu_gcn.npz"))
CASES = {
    "value48": ((8, 6), 7, 3, dict(gc_model="stgcn", graph_node_format="node_per_value", num_layers=5, inner_feature_dim=16)),
    "sensor16": ((8, 6), 5, 4, dict(gc_model="stgcn", graph_node_format="node_per_sensor", num_signals=2, num_layers=4,
                                    inner_feature_dim=8, adjacency_normalization="row", num_temporal_back_connections=2,
                                    inter_signal_back_connections=True)),
}


def build(tag, shape=None, classes=None, kw=None, double=False):
    """-> (model with filler parameters, the same values as a float64 state dict under the reference's key names).
    ``double``: fill in float64 (the golden vectors' precision) instead of float32 (what the GPU model holds)."""
    from fusion_gcn_amd.models.mmargcn.mmargcn import Model
    if shape is None:
        shape, classes, _, kw = CASES[tag]
    model = Model({"inertial": shape}, classes, None, mode="imu_gcn", **kw)
    if double:
        model = model.double()
    filler.fill_state_dict(model.state_dict(), skip=("adj",), rename=lambda k: k.replace("_model.", ""))
    sd = {k.replace("_model.", ""): (v.detach().double().clone() if v.is_floating_point() else v.detach().clone())
          for k, v in model.state_dict().items()}
    return model, sd


def inputs(tag, shape, batch, classes):
    x = torch.from_numpy(filler.bellish(f"x.{tag}", (batch, *shape), scale=0.5)).double()
    y = torch.from_numpy(filler.uniform(f"y.{tag}", (batch,), 0, classes).astype(np.int64))
    return x, y


def fmt(kw):
    return dict(graph_node_format=kw["graph_node_format"],
                num_features=1 if kw["graph_node_format"] == "node_per_value" else 6 // kw["num_signals"])


def test_imu_graph_adjacency_matches_the_reference():
    from fusion_gcn_amd.models.mmargcn.imu_feature_models import build_imu_graph_adjacency
    for tag, (shape, sig, kw) in {"row_t1": ((5, 3), 0, dict(normalization="row")),
                                  "column_t2_inter": ((5, 3), 0, dict(normalization="column", temporal_back_connections=2,
                                                                       inter_signal_back_connections=True)),
                                  "symmetric_sensor": ((6, 4), 2, dict(normalization="symmetric"))}.items():
        want = GOLD[f"adj.{tag}"]
        got = build_imu_graph_adjacency(shape, sig, "stgcn", False, **kw).double().numpy()
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-7          # (the reference stores float32)
        edges, n = O.imu_graph_edges(shape, sig, kw.get("temporal_back_connections", 1),
                                     kw.get("inter_signal_back_connections", False))
        mine = O.normalized_adjacency(edges, n, kw["normalization"], True)
        assert np.abs(mine - want).max() < 1e-7


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_and_state_dict_surface_match_the_reference(tag):
    shape, classes, batch, kw = CASES[tag]
    model, sd = build(tag, double=True)
    assert sorted(sd) == list(GOLD[f"{tag}.keys"])                                  # same keys as the reference's state dict
    assert np.abs(sd["gcn.gc1.adj"].numpy() - GOLD[f"{tag}.adj"]).max() < 1e-7
    x, y = inputs(tag, shape, batch, classes)
    assert np.array_equal(y.numpy(), GOLD[f"{tag}.labels"])
    assert rel_l2(O.imu_gcn_forward(x, sd, train=False, **fmt(kw)).numpy(), GOLD[f"{tag}.eval.logits"]) < 1e-10
    logits, loss, grads = O.loss_and_grads(x, y, sd, **fmt(kw))
    assert rel_l2(logits.numpy(), GOLD[f"{tag}.train.logits"]) < 1e-10
    assert abs(float(loss) - float(GOLD[f"{tag}.train.loss"])) < 1e-10
    for k, g in grads.items():
        want = GOLD[f"{tag}.grad.{k}"]
        if np.abs(want).max() < 1e-12:
            assert g is None or float(g.abs().max()) < 1e-12
        else:
            assert rel_l2(g.numpy(), want) < 1e-9, k


def test_agcn_variant_and_other_modes_fail_loudly():
    from fusion_gcn_amd.models.mmargcn.mmargcn import Model
    with pytest.raises(NotImplementedError):
        Model({"inertial": (8, 6)}, 5, None, mode="imu_gcn", gc_model="agcn", graph_node_format="node_per_sensor", num_signals=2)
    with pytest.raises(NotImplementedError):
        Model({"inertial": (8, 6)}, 5, None, mode="imu_signal_image")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["value48", "sensor16", "value240", "value1956", "wide2048"])
def test_hip_imu_gcn_matches_the_oracle(tag):
    dev = torch.device("cuda:0")
    if tag in CASES:
        shape, classes, batch, kw = CASES[tag]
    elif tag == "value240":   # V = 240 nodes (not a multiple of 64: padded contraction), all three residual kinds, widths to 128
        shape, classes, batch = (40, 6), 27, 4
        kw = dict(gc_model="stgcn", graph_node_format="node_per_value", num_layers=5, inner_feature_dim=64)
    elif tag == "wide2048":   # the config's widths: a Conv1d + BatchNorm residual into 2048 channels (1024-channel windows)
        shape, classes, batch = (8, 6), 27, 2
        kw = dict(gc_model="stgcn", graph_node_format="node_per_value", num_layers=4, inner_feature_dim=1024)
    else:                     # the UTD-MHAD config's node count (326 x 6 = 1956), narrow
        shape, classes, batch = (326, 6), 27, 2
        kw = dict(gc_model="stgcn", graph_node_format="node_per_value", num_layers=3, inner_feature_dim=32)
    model, sd = build(tag, shape, classes, kw)
    x, y = inputs(tag, shape, batch, classes)
    ref_eval = O.imu_gcn_forward(x, sd, train=False, **fmt(kw))
    ref_logits, ref_loss, ref_grads = O.loss_and_grads(x, y, sd, **fmt(kw))
    model = model.to(dev)
    model.eval()
    with torch.no_grad():
        got_eval = model(x.float().to(dev)).cpu().double()
    assert rel_l2(got_eval.numpy(), ref_eval.numpy()) < 2e-5
    model.train()
    logits = model(x.float().to(dev))
    loss = F.cross_entropy(logits, y.to(dev))
    loss.backward()
    assert rel_l2(logits.detach().cpu().double().numpy(), ref_logits.numpy()) < 2e-5
    assert abs(float(loss.detach()) - float(ref_loss)) < 2e-5 * max(1.0, abs(float(ref_loss)))
    scale = max(float(g.abs().max()) for g in ref_grads.values() if g is not None)
    for name, p in model.named_parameters():
        k = name.replace("_model.", "")
        want = ref_grads[k]
        got = p.grad.detach().cpu().double()
        if k.endswith("residual.0.bias"):                # in front of a train-mode BatchNorm: exactly zero here
            assert float(got.abs().max()) == 0.0 and float(want.abs().max()) < 1e-9 * max(1.0, scale)
            continue
        assert rel_l2(got.numpy(), want.numpy()) < 5e-4, (k, rel_l2(got.numpy(), want.numpy()))
    if tag in CASES:                                      # and against the reference's own numbers
        assert rel_l2(logits.detach().cpu().double().numpy(), GOLD[f"{tag}.train.logits"]) < 2e-5
        bn = dict(model.named_buffers())
        for k in GOLD.files:
            if k.startswith(f"{tag}.after."):
                name = "_model." + k[len(f"{tag}.after."):]
                assert rel_l2(bn[name].cpu().double().numpy(), GOLD[k]) < 1e-5, k
