"""Pin the CPU oracle (oracle/agcn_oracle.py) to the reference: every case compares the restatement, run in
float64 on filler-generated parameters/inputs, with outputs the imported reference produced for the same
parameters/inputs (tests/golden/*.npz, written by oracle/gen_golden.py).  Tolerance 1e-10 relative: both
sides are float64 torch, only summation order differs."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from fusion_gcn_amd.datasets.mmact import constants as mmact
from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
from fusion_gcn_amd.datasets.utd_mhad import constants as utd
from oracle import agcn_oracle as O
from oracle import filler, graph_oracle

TOL = 1e-10


def t64(a):
    return torch.from_numpy(np.ascontiguousarray(a)).double()


def adj_for(name):
    if name == "ntu":
        return graph_oracle.spatial_partition_stack(ntu.skeleton_edges)
    if name == "mmact":
        return graph_oracle.spatial_partition_stack(mmact.skeleton_edges)
    if name == "utd":
        return graph_oracle.spatial_partition_stack(utd.skeleton_edges)
    if name == "utd_imu2_center":
        return graph_oracle.spatial_partition_stack(
            graph_oracle.imu_fusion_edges(utd.skeleton_edges, 20, "append_center", 2, center_joint=1))
    if name == "ntu_imu2_center":
        return graph_oracle.spatial_partition_stack(
            graph_oracle.imu_fusion_edges(ntu.skeleton_edges, 25, "append_center", 2, center_joint=ntu.center_joint))
    if name == "mmact_imu4_center":
        return graph_oracle.spatial_partition_stack(
            graph_oracle.imu_fusion_edges(mmact.skeleton_edges, 18, "append_center", 4, center_joint=mmact.center_joint))
    raise KeyError(name)


def filled(keys_shapes, dtype=torch.float64):
    return {k: torch.from_numpy(np.ascontiguousarray(filler.fill_value_for(k, s))).reshape(s).to(
        torch.long if k.endswith("num_batches_tracked") else dtype) for k, s in keys_shapes.items()}


def sub_state(prefix, cin, cout, adj, kinds):
    """State dict of one block (keys under ``prefix``) built through the oracle's own constructor."""
    v = adj.shape[1]
    full = O.new_state_dict((1, 8, v, cin), 5, adj, num_layers=1, start=cout, dtype=torch.float64)
    out = {}
    for k, t in full.items():
        if k.startswith(prefix):
            out[k] = t
    return out


def fill_block(sd):
    for k in list(sd):
        if k.endswith("adj_a"):
            continue
        sd[k] = torch.from_numpy(np.ascontiguousarray(filler.fill_value_for(k, tuple(sd[k].shape)))).reshape(
            sd[k].shape).to(sd[k].dtype)
    return sd


def grads_probe(out, tag, tensors):
    w = t64(filler.uniform(f"probe.{tag}", tuple(out.shape), -1.0, 1.0))
    return torch.autograd.grad((out * w).sum(), tensors, allow_unused=True)


def check_module_case(ref, tag, sd, prefix, x, fwd):
    """train fwd + grads + running stats, eval fwd — against the stored reference outputs."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "adj_a"))}
    live = dict(sd)
    live.update(params)
    xin = x.clone().requires_grad_(True)
    stats = O.Stats()
    out = fwd(xin, live, True, stats)
    assert rel_l2(out.detach().numpy(), ref[f"{tag}.train.out"]) < TOL
    gs = grads_probe(out, tag, [xin] + list(params.values()))
    assert rel_l2(gs[0].numpy(), ref[f"{tag}.train.dx"]) < TOL
    for (k, _), g in zip(params.items(), gs[1:]):
        name = k[len(prefix):]
        want = ref[f"{tag}.train.grad.{name}"]
        got = np.zeros_like(want) if g is None else g.numpy()
        # analytically-zero grads (biases in front of a train-mode BN, conv_a bias) are ~1e-15 noise on
        # both sides: absolute floor next to the relative bound
        assert np.linalg.norm(got - want) <= 1e-8 * np.linalg.norm(want) + 1e-11, (tag, name)
    for k, v in stats.updates.items():
        if k.endswith(("running_mean", "running_var")):
            assert rel_l2(v.numpy(), ref[f"{tag}.train.after.{k[len(prefix):]}"]) < TOL, k
    out_eval = fwd(x, sd, False, None)
    assert rel_l2(out_eval.numpy(), ref[f"{tag}.eval.out"]) < TOL


@pytest.mark.parametrize("tag,cin,cout,gname", [("sgc_3_16_ntu", 3, 16, "ntu"), ("sgc_16_16_mmact", 16, 16, "mmact"),
                                                ("sgc_16_32_utd22", 16, 32, "utd_imu2_center")])
def test_spatial_graph_conv(golden, tag, cin, cout, gname):
    ref = golden("spatial_graph_conv.npz")
    adj = adj_for(gname)
    sd = fill_block(sub_state("l0.gcn1.", cin, cout, adj, None))
    x = t64(filler.bellish(f"x.{tag}", (2, cin, 6, adj.shape[1])))
    check_module_case(ref, tag, sd, "l0.gcn1.", x,
                      lambda xi, s, tr, st: O.spatial_graph_conv(xi, s, "l0.gcn1", tr, st)[0])
    _, adj_c = O.spatial_graph_conv(x, sd, "l0.gcn1", False)
    got = torch.stack(adj_c).numpy()
    assert rel_l2(got, ref[f"{tag}.adj_c"]) < TOL
    np.testing.assert_allclose(got.sum(axis=-2), 1.0, atol=1e-12)      # softmax over dim -2: columns sum to 1


@pytest.mark.parametrize("tag,cin,cout,k,s", [("tc_k9_s1", 16, 16, 9, 1), ("tc_k9_s2", 16, 16, 9, 2),
                                              ("tc_k1_s2", 8, 16, 1, 2)])
def test_temporal_conv(golden, tag, cin, cout, k, s):
    ref = golden("temporal_conv.npz")
    shapes = {"l0.tcn1.conv.weight": (cout, cin, k, 1), "l0.tcn1.conv.bias": (cout,),
              "l0.tcn1.bn.weight": (cout,), "l0.tcn1.bn.bias": (cout,), "l0.tcn1.bn.running_mean": (cout,),
              "l0.tcn1.bn.running_var": (cout,), "l0.tcn1.bn.num_batches_tracked": ()}
    sd = filled(shapes)
    x = t64(filler.bellish(f"x.{tag}", (2, cin, 11, 5)))
    check_module_case(ref, tag, sd, "l0.tcn1.", x, lambda xi, s_, tr, st: O.temporal_conv(xi, s_, "l0.tcn1", s, tr, st))
    # output length (T-1)//s + 1
    assert O.temporal_conv(x, sd, "l0.tcn1", s, False).shape[2] == (11 - 1) // s + 1


@pytest.mark.parametrize("tag,cin,cout,s,res", [("stc_first", 3, 16, 1, False), ("stc_identity", 16, 16, 1, True),
                                                ("stc_down_s2", 16, 32, 2, True)])
def test_st_block(golden, tag, cin, cout, s, res):
    ref = golden("st_block.npz")
    adj = adj_for("ntu")
    full = O.new_state_dict((1, 8, 25, cin), 5, adj, num_layers=1, start=cout, dtype=torch.float64)
    sd = {k: v for k, v in full.items() if k.startswith("l0.")}
    if res and not (cin == cout and s == 1):
        sd.update({"l0.residual.conv.weight": torch.zeros(cout, cin, 1, 1, dtype=torch.float64),
                   "l0.residual.conv.bias": torch.zeros(cout, dtype=torch.float64)})
        sd.update({k.replace("tcn1.bn", "residual.bn"): v.clone() for k, v in sd.items() if ".tcn1.bn." in k})
    sd = fill_block(sd)
    x = t64(filler.bellish(f"x.{tag}", (2, cin, 10, 25)))
    check_module_case(ref, tag, sd, "l0.", x, lambda xi, s_, tr, st: O.st_block(xi, s_, "l0", s, res, tr, st)[0])


@pytest.mark.parametrize("tag,shape,gname,classes", [("cfg1", (2, 1, 100, 20, 3), "utd", 27),
                                                     ("cfg2_small", (2, 2, 32, 25, 3), "ntu", 60)])
def test_full_model(golden, tag, shape, gname, classes):
    ref = golden("model.npz")
    n, m, t, v, c = shape
    sd = O.new_state_dict((m, t, v, c), classes, adj_for(gname), dtype=torch.float64)
    sd = fill_block(sd)
    x = t64(filler.skeleton_input(f"x.{tag}", shape, empty_second_body=(m > 1)))
    labels = torch.from_numpy(filler.uniform(f"y.{tag}", (n,), 0, classes).astype(np.int64))
    np.testing.assert_array_equal(labels.numpy(), ref[f"{tag}.labels"])
    assert rel_l2(O.model_forward(x, sd, train=False).numpy(), ref[f"{tag}.eval.logits"]) < TOL
    logits, loss, grads, stats = O.loss_and_grads(x, labels, sd)
    assert rel_l2(logits.numpy(), ref[f"{tag}.train.logits"]) < TOL
    assert abs(float(loss) - float(ref[f"{tag}.train.loss"])) < 1e-11
    flat_err, flat_ref = 0.0, 0.0
    for k, g in grads.items():
        want_l2 = float(ref[f"{tag}.gl2.{k}"])
        want_sum = float(ref[f"{tag}.gsum.{k}"])
        tol = 1e-8 * want_l2 + 1e-11
        assert abs(float(g.norm()) - want_l2) <= tol, k
        assert abs(float(g.sum()) - want_sum) <= tol * max(1.0, g.numel() ** 0.5), k
        if f"{tag}.grad.{k}" in ref.files:
            assert np.linalg.norm(g.numpy() - ref[f"{tag}.grad.{k}"]) <= tol, k
    for k, vv in stats.updates.items():
        if f"{tag}.after.{k}" in ref.files:
            assert rel_l2(vv.numpy(), ref[f"{tag}.after.{k}"]) < TOL, k
    # the float32 oracle agrees with the float32 reference at the fp32 noise floor
    sd32 = {k: (vv.float() if vv.is_floating_point() else vv) for k, vv in sd.items()}
    lg32 = O.model_forward(x.float(), sd32, train=True)
    assert rel_l2(lg32.numpy(), ref[f"{tag}.train.logits_f32"]) < 1e-5


def test_mmargcn_spatial_fusion_logits(golden):
    ref = golden("mmargcn.npz")
    shape = (2, 1, 20, 22, 3)
    sd = fill_block(O.new_state_dict(shape[1:], 27, adj_for("utd_imu2_center"), dtype=torch.float64))
    x = t64(filler.skeleton_input("x.mm22", shape))
    assert rel_l2(O.model_forward(x, sd, train=False).numpy(), ref["mm22.eval.logits"]) < TOL
    assert rel_l2(O.model_forward(x, sd, train=True).numpy(), ref["mm22.train.logits"]) < TOL


@pytest.mark.parametrize("tag,gname,shape,classes", [("ntu27", "ntu_imu2_center", (2, 2, 16, 27, 3), 60),
                                                     ("mmact22", "mmact_imu4_center", (2, 2, 16, 22, 3), 35),
                                                     ("mmact18", "mmact", (2, 2, 16, 18, 2), 35)])
def test_other_baseline_shapes(golden, tag, gname, shape, classes):
    """BASELINE configs 3 and 4 at fixture size (NTU graph + 2 IMU joints, V = 27; MMAct COCO-18 + 4 IMU joints, V = 22,
    35 classes; MMAct skeleton-only, C = 2): logits, loss and every parameter-gradient norm against the reference."""
    ref = golden("mmargcn.npz")
    sd = fill_block(O.new_state_dict(shape[1:], classes, adj_for(gname), dtype=torch.float64))
    x = t64(filler.skeleton_input(f"x.{tag}", shape, empty_second_body=True))
    labels = torch.from_numpy(ref[f"{tag}.labels"])
    assert rel_l2(O.model_forward(x, sd, train=False).numpy(), ref[f"{tag}.eval.logits"]) < TOL
    logits, loss, grads, _ = O.loss_and_grads(x, labels, sd)
    assert rel_l2(logits.numpy(), ref[f"{tag}.train.logits"]) < TOL
    assert abs(float(loss) - float(ref[f"{tag}.train.loss"])) < 1e-10
    for k, g in grads.items():
        key = f"{tag}.gl2.{k}" if f"{tag}.gl2.{k}" in ref.files else f"{tag}.gl2._model.agcn.{k}"   # mmargcn wraps the AGCN
        want = float(ref[key])
        assert abs(float(g.norm()) - want) <= 1e-8 * want + 1e-11, k


def test_state_dict_manifest_matches_reference():
    with open(os.path.join(GOLDEN, "manifests.json")) as f:
        man = json.load(f)
    sd = O.new_state_dict((2, 300, 25, 3), 60, adj_for("ntu"))
    assert {k: list(v.shape) for k, v in sd.items()} == man["mmargcn.agcn"]
    sd = O.new_state_dict((1, 100, 20, 3), 27, adj_for("utd"))
    mapped = {O.agcn_key(k): list(v.shape) for k, v in sd.items() if not k.endswith("adj_a")}
    assert mapped == man["agcn"]
    assert len(man["agcn"]) == 352


def test_filler_is_stable():
    """The filler is part of the fixture contract: pin a few values so a refactor cannot silently drift."""
    u = filler.uniform("pin", (4,), -1, 1)
    v = filler.fill_value_for("l3.gcn1.conv_d.1.weight", (2, 3, 1, 1))
    assert u.shape == (4,) and np.all(np.abs(u) <= 1)
    np.testing.assert_allclose(u, filler.uniform("pin", (4,), -1, 1))
    assert not np.allclose(v, filler.fill_value_for("l3.gcn1.conv_d.2.weight", (2, 3, 1, 1)))
    np.testing.assert_allclose(float(u[0]), PIN_U0, rtol=0, atol=1e-15)


PIN_U0 = float(filler.uniform("pin", (4,), -1, 1)[0])
