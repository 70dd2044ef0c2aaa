#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generate tests/golden/* by IMPORTING THE REFERENCE (build container only).

Run from anywhere:  python oracle/gen_golden.py
Needs /root/reference (read-only mount); nothing is written there.  Import recipe = SURVEY.md Appendix B:
stub torchvision / cv2, restore np.int / np.float, PYTHONDONTWRITEBYTECODE.  The reference modules are
constructed, their state dicts are filled by oracle.filler (deterministic, build-owned), inputs come from
the same filler, and only inputs-by-name + OUTPUTS are stored (float64 .npz, a few hundred KB in total).
The GPU box never sees the reference: tests read only the .npz/.json written here.
"""
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np  # noqa: E402

np.int = int      # reference util/graph.py:75,88,117,127 use the removed aliases
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

from util.graph import Graph  # noqa: E402  (reference)
from util.partition_strategy import GraphPartitionStrategy  # noqa: E402  (reference)
import models.mmargcn.agcn as ref_agcn  # noqa: E402  (reference)
import models.mmargcn.mmargcn as ref_mm  # noqa: E402  (reference)
import models.agcn.agcn as ref_agcn_cuda  # noqa: E402  (reference; constructed for its key manifest only)
from models.mmargcn.fusion import get_skeleton_imu_fusion_graph  # noqa: E402  (reference)
from datasets.utd_mhad import constants as utd  # noqa: E402  (reference)
from datasets.mmact import constants as mmact  # noqa: E402  (reference)
from datasets.ntu_rgb_d import constants as ntu  # noqa: E402  (reference)

from oracle import filler  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)
torch.set_num_threads(8)


def graphs():
    g = {}
    base = {"utd": (utd, {}), "mmact": (mmact, {}), "ntu": (ntu, {})}
    for name, (c, _) in base.items():
        g[name] = Graph(c.skeleton_edges, center_joint=c.center_joint)
    g["utd_imu2_center"] = get_skeleton_imu_fusion_graph(g["utd"], "append_center", 2)
    g["utd_imu2_center_ic"] = get_skeleton_imu_fusion_graph(g["utd"], "append_center", 2, interconnect_imu_joints=True)
    g["utd_imu2_right"] = get_skeleton_imu_fusion_graph(g["utd"], "append_right", 2, right_wrist_joint=10, right_hip_joint=16)
    g["utd_imu2_right_ic"] = get_skeleton_imu_fusion_graph(g["utd"], "append_right", 2, right_wrist_joint=10,
                                                           right_hip_joint=16, interconnect_imu_joints=True)
    g["mmact_imu4_center"] = get_skeleton_imu_fusion_graph(g["mmact"], "append_center", 4)
    g["mmact_imu4_center_ic"] = get_skeleton_imu_fusion_graph(g["mmact"], "append_center", 4, interconnect_imu_joints=True)
    g["ntu_imu2_center"] = get_skeleton_imu_fusion_graph(g["ntu"], "append_center", 2)
    return g


def adjacency(g):
    return GraphPartitionStrategy().get_adjacency_matrix_array(g)


def t64(a):
    return torch.from_numpy(np.ascontiguousarray(a)).double()


def grads_of(out, weight_name, tensors):
    """d/d(tensors) of sum(out * W) with W a filler tensor — a deterministic scalar probe."""
    w = t64(filler.uniform(weight_name, tuple(out.shape), -1.0, 1.0))
    return torch.autograd.grad((out * w).sum(), tensors, allow_unused=True)


def module_case(mod, prefix, x, tag, store, fwd=lambda m, x: m(x)):
    mod.double()
    filler.fill_state_dict(mod.state_dict(), prefix=prefix)
    for ev in (False, True):
        mod.train(not ev)
        xin = x.clone().requires_grad_(not ev)
        sd_before = {k: v.clone() for k, v in mod.state_dict().items()}
        out = fwd(mod, xin)
        key = f"{tag}.{'eval' if ev else 'train'}"
        store[f"{key}.out"] = out.detach().numpy()
        if not ev:
            params = dict(mod.named_parameters())
            gs = grads_of(out, f"probe.{tag}", [xin] + list(params.values()))
            store[f"{key}.dx"] = gs[0].numpy()
            for (n, _), gr in zip(params.items(), gs[1:]):
                store[f"{key}.grad.{n}"] = (torch.zeros(()) if gr is None else gr).numpy()
            for k, v in mod.state_dict().items():
                if k.endswith(("running_mean", "running_var")):
                    store[f"{key}.after.{k}"] = v.numpy().copy()
            # restore running stats so eval uses the filled ones
            mod.load_state_dict(sd_before)
    return store


def relu_sign_taps(model):
    """Forward hooks on the reference Model's blocks: [out > 0] of every gcn1 (G) and block (O) output, in network order; the
    list fills during the next forward."""
    signs = []
    for i in range(10):
        blk = getattr(model, f"l{i}")
        blk.gcn1.register_forward_hook(lambda _m, _i, out: signs.append((out.detach() > 0)))
        blk.register_forward_hook(lambda _m, _i, out: signs.append((out.detach() > 0)))
    return signs


def main():
    gs = graphs()
    # ---- (i) adjacency stacks -------------------------------------------------------------------
    np.savez(os.path.join(OUT, "adjacency.npz"), **{k: adjacency(g) for k, g in gs.items()},
             **{f"edges.{k}": g.edges for k, g in gs.items()})

    # ---- (ii) SpatialGraphConv ------------------------------------------------------------------
    store = {}
    for tag, cin, cout, gname in (("sgc_3_16_ntu", 3, 16, "ntu"), ("sgc_16_16_mmact", 16, 16, "mmact"),
                                  ("sgc_16_32_utd22", 16, 32, "utd_imu2_center")):
        adj = adjacency(gs[gname])
        v = adj.shape[1]
        mod = ref_agcn.SpatialGraphConv(cin, cout, adj)
        x = t64(filler.bellish(f"x.{tag}", (2, cin, 6, v)))
        module_case(mod, "l0.gcn1.", x, tag, store)
        mod.eval()
        mod(x)
        store[f"{tag}.adj_c"] = torch.stack([a.detach() for a in mod.adj_c]).numpy()
    np.savez(os.path.join(OUT, "spatial_graph_conv.npz"), **store)

    # ---- (iii) TemporalConv ---------------------------------------------------------------------
    store = {}
    for tag, cin, cout, k, s in (("tc_k9_s1", 16, 16, 9, 1), ("tc_k9_s2", 16, 16, 9, 2), ("tc_k1_s2", 8, 16, 1, 2)):
        mod = ref_agcn.TemporalConv(cin, cout, kernel_size=k, stride=s)
        x = t64(filler.bellish(f"x.{tag}", (2, cin, 11, 5)))
        module_case(mod, "l0.tcn1.", x, tag, store)
    np.savez(os.path.join(OUT, "temporal_conv.npz"), **store)

    # ---- (iv) SpatialTemporalConv ---------------------------------------------------------------
    store = {}
    adj = adjacency(gs["ntu"])
    for tag, cin, cout, s, res in (("stc_first", 3, 16, 1, False), ("stc_identity", 16, 16, 1, True),
                                   ("stc_down_s2", 16, 32, 2, True)):
        mod = ref_agcn.SpatialTemporalConv(cin, cout, adj, stride=s, residual=res)
        x = t64(filler.bellish(f"x.{tag}", (2, cin, 10, 25)))
        module_case(mod, "l0.", x, tag, store)
    np.savez(os.path.join(OUT, "st_block.npz"), **store)

    # ---- (v) full Model, cfg-1 shape (N=2, M=1, T=100, V=20, C=3; 27 classes; UTD graph) --------
    store = {}
    for tag, shape, gname, classes in (("cfg1", (2, 1, 100, 20, 3), "utd", 27),
                                       ("cfg2_small", (2, 2, 32, 25, 3), "ntu", 60)):
        n, m, t, v, c = shape
        model = ref_agcn.Model((m, t, v, c), classes, gs[gname]).double()
        filler.fill_state_dict(model.state_dict())
        x = t64(filler.skeleton_input(f"x.{tag}", shape, empty_second_body=(m > 1)))
        labels = torch.from_numpy((filler.uniform(f"y.{tag}", (n,), 0, classes)).astype(np.int64))
        store[f"{tag}.labels"] = labels.numpy()
        before = {k: v.clone() for k, v in model.state_dict().items()}
        model.eval()
        store[f"{tag}.eval.logits"] = model(x).detach().numpy()
        model.train()
        signs64 = relu_sign_taps(model)
        logits = model(x)
        loss = F.cross_entropy(logits, labels)
        loss.backward()
        store[f"{tag}.train.logits"] = logits.detach().numpy()
        store[f"{tag}.train.loss"] = loss.detach().numpy()
        for name, p in model.named_parameters():
            g = p.grad
            store[f"{tag}.gsum.{name}"] = g.sum().numpy()
            store[f"{tag}.gl2.{name}"] = g.norm().numpy()
            if p.numel() <= 4096:
                store[f"{tag}.grad.{name}"] = g.numpy()
        for k, v_ in model.state_dict().items():
            if k.endswith(("running_mean", "running_var")) and k.split(".")[0] in ("data_bn", "l0", "l4", "l9"):
                store[f"{tag}.after.{k}"] = v_.numpy().copy()
        # fp32 run of the reference itself: the noise floor the fp32 HIP path is judged against
        model32 = ref_agcn.Model((m, t, v, c), classes, gs[gname])
        model32.load_state_dict({k: (v_.float() if v_.is_floating_point() else v_) for k, v_ in before.items()})
        model32.train()
        signs32 = relu_sign_taps(model32)
        lg32 = model32(x.float())
        F.cross_entropy(lg32, labels).backward()
        store[f"{tag}.train.logits_f32"] = lg32.detach().numpy()
        flat64 = torch.cat([p.grad.flatten() for p in model.parameters()])
        flat32 = torch.cat([p.grad.flatten() for p in model32.parameters()]).double()
        store[f"{tag}.ref_f32_vs_f64_grad_rel"] = ((flat32 - flat64).norm() / flat64.norm()).numpy()
        # ReLU decisions (the 20 outputs G = relu(gcn) and O = relu(tcn + residual), agcn.py:113-115,135-136) on which the
        # reference's own float32 run differs from its float64 run: the yardstick for the HIP forward's flip count
        store[f"{tag}.ref_f32_vs_f64_relu_flips"] = np.array([int((a != b).sum()) for a, b in zip(signs64, signs32)], dtype=np.int64)
        store[f"{tag}.relu_decisions"] = np.array([a.numel() for a in signs64], dtype=np.int64)
    np.savez(os.path.join(OUT, "model.npz"), **store)

    # ---- (vi) mmargcn.Model(mode="skeleton_imu_spatial_fusion"), V = 20 + 2 ---------------------
    store = {}
    shape = (2, 1, 20, 22, 3)
    mm = ref_mm.Model({"skeleton": shape[1:]}, 27, gs["utd"], mode="skeleton_imu_spatial_fusion",
                      num_imu_joints=2, imu_enhanced_mode="append_center").double()
    filler.fill_state_dict(mm.state_dict(), rename=lambda k: k.replace("_model.agcn.", ""))
    x = t64(filler.skeleton_input("x.mm22", shape))
    mm.eval()
    store["mm22.eval.logits"] = mm(x).detach().numpy()
    mm.train()
    store["mm22.train.logits"] = mm(x).detach().numpy()
    # ---- (vi-b) the other BASELINE shapes at fixture size: cfg 3 on the NTU graph (V = 25 + 2, M = 2), cfg 4 = MMAct
    #      COCO-18 graph + 4 IMU joints (V = 22, C = 3, M = 2, 35 classes) and MMAct skeleton-only (V = 18, C = 2)
    def model_case(tag, model, shape, classes):
        n = shape[0]
        x = t64(filler.skeleton_input(f"x.{tag}", shape, empty_second_body=(shape[1] > 1)))
        labels = torch.from_numpy((filler.uniform(f"y.{tag}", (n,), 0, classes)).astype(np.int64))
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
            store[f"{tag}.gl2.{name}"] = p.grad.norm().numpy()

    for tag, gname, shape, classes, n_imu in (("ntu27", "ntu", (2, 2, 16, 27, 3), 60, 2),
                                              ("mmact22", "mmact", (2, 2, 16, 22, 3), 35, 4)):
        m2 = ref_mm.Model({"skeleton": shape[1:]}, classes, gs[gname], mode="skeleton_imu_spatial_fusion",
                          num_imu_joints=n_imu, imu_enhanced_mode="append_center").double()
        filler.fill_state_dict(m2.state_dict(), rename=lambda k: k.replace("_model.agcn.", ""))
        model_case(tag, m2, shape, classes)
    m3 = ref_agcn.Model((2, 16, 18, 2), 35, gs["mmact"]).double()
    filler.fill_state_dict(m3.state_dict())
    model_case("mmact18", m3, (2, 2, 16, 18, 2), 35)
    np.savez(os.path.join(OUT, "mmargcn.npz"), **store)

    # ---- (vii) state-dict manifests -------------------------------------------------------------
    man = {}
    m_agcn = ref_agcn_cuda.Model({"skeleton": (1, 100, 20, 3)}, 27, gs["utd"])
    man["agcn"] = {k: list(v.shape) for k, v in m_agcn.state_dict().items()}
    man["mmargcn.skeleton_imu_spatial_fusion"] = {k: list(v.shape) for k, v in mm.state_dict().items()}
    man["mmargcn.agcn"] = {k: list(v.shape) for k, v in ref_agcn.Model((2, 300, 25, 3), 60, gs["ntu"]).state_dict().items()}
    man["torch_version"] = torch.__version__
    with open(os.path.join(OUT, "manifests.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    sizes = {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))}
    print(json.dumps(sizes, indent=1))


if __name__ == "__main__":
    main()
