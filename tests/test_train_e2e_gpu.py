"""End to end: feature file -> ClipBatches -> HIP AGCN model -> FlatOptimizer, three training steps, against the same loop on
the CPU oracle (float64 model + torch.optim.SGD, the reference's own optimizer class, session_helper.py:48-53).

SGD rather than the configs' Adam on purpose: in train mode the conv biases in front of a BatchNorm have an analytically zero
gradient, which the HIP path returns as exact zeros and autograd as ~1e-17 rounding noise; Adam normalises that noise into
+-lr steps on parameters the output does not depend on, so parameter trajectories under Adam are not comparable (losses and
logits are).  The Adam arithmetic itself is pinned in tests/test_optim.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import agcn_oracle as O
from oracle import filler

pytestmark = pytest.mark.gpu


def test_three_training_steps_match_the_oracle_loop(tmp_path, fgcn_math):
    from fusion_gcn_amd.data import ClipBatches, MultiModalDataset, NumpyDatasetLoader, NumpyWriter
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.optim import FlatOptimizer
    from fusion_gcn_amd.util import Graph
    dev = torch.device("cuda:0")
    n, shape, classes, bs = 12, (1, 24, 20, 3), 27, 4
    feats = filler.skeleton_input("x.e2e", (n, *shape)).astype(np.float32)
    labels = filler.uniform("y.e2e", (n,), 0, classes).astype(np.int64)
    with NumpyWriter(str(tmp_path / "skeleton_train_features.npy"), np.float32, feats.shape) as w:
        for s in feats:
            w.collect_next(s)
    np.save(tmp_path / "train_labels.npy", labels)
    ds = MultiModalDataset([(str(tmp_path), NumpyDatasetLoader())], "train")
    assert tuple(ds.get_input_shape()["skeleton"]) == shape and ds.get_num_classes() <= classes

    model = Model(shape, classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint))
    filler.fill_state_dict(model.state_dict())
    sd = {k: (v.detach().double().clone() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    hyper = dict(momentum=0.9, nesterov=True, weight_decay=1e-4)
    opt = FlatOptimizer(model.parameters(), "SGD", 0.01, **hyper)
    names = [k for k, _ in model.named_parameters()]
    ref_params = [sd[k].requires_grad_(False) for k in names]
    ref_opt = torch.optim.SGD(ref_params, 0.01, **hyper)

    batches = ClipBatches(ds, bs, shuffle=False, drop_last=True, device=dev, resident=False)      # the pinned streaming path
    losses, ref_losses = [], []
    for x, y, idx in batches:
        opt.zero_grad()
        loss = F.cross_entropy(model(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        xi = torch.from_numpy(feats[idx.numpy()]).double()
        _, ref_loss, grads, _ = O.loss_and_grads(xi, torch.from_numpy(labels[idx.numpy()]), sd)
        ref_opt.zero_grad()
        for k, p in zip(names, ref_params):
            p.grad = grads[k]
        ref_opt.step()
        ref_losses.append(float(ref_loss))
    assert len(losses) == 3
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 2e-4 * max(1.0, abs(b)), (losses, ref_losses)
    # the trained parameters: every tensor within 2e-3 of the oracle loop's (north star: 1e-3 rel per step, three steps)
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        want = sd[k].numpy()
        if np.abs(want).max() == 0:
            continue
        err = rel_l2(p.detach().cpu().double().numpy(), want)
        worst = max(worst, (k, err), key=lambda t: t[1])
    assert worst[1] < 2e-3, worst
    # and what they compute: logits of a held-out pass in eval mode (running statistics were updated on both sides)
    xe = torch.from_numpy(feats[:bs])
    model.eval()
    with torch.no_grad():
        got = model(xe.to(dev)).cpu().double()
    # the oracle's forward does not write running statistics back: compare in train mode instead (batch statistics)
    model.train()
    with torch.no_grad():
        got_t = model(xe.to(dev)).cpu().double()
    want_t = O.model_forward(xe.double(), sd, train=True)
    # (three updates amplify the fp32-vs-fp64 ReLU-flip floor of the gradients, 3e-4..2e-3 per step: SURVEY.md section 0 fact 9)
    assert rel_l2(got_t.numpy(), want_t.detach().numpy()) < 5e-3
    assert torch.isfinite(got).all()


def test_late_fusion_trains_from_two_feature_files(tmp_path, fgcn_math):
    """Two modalities on disk -> MultiModalDataset / ClipBatches (dict batches) -> mode skeleton_imu_gcn_late_fusion with the AGCN
    IMU branch -> FlatOptimizer: two SGD steps against the same loop on the float64 oracle."""
    import test_imu_gcn as TI
    from fusion_gcn_amd.data import ClipBatches, MultiModalDataset, NumpyDatasetLoader, NumpyWriter
    from fusion_gcn_amd.optim import FlatOptimizer
    dev = torch.device("cuda:0")
    n, bs = 6, 3
    feats = {"skeleton": filler.skeleton_input("x.e2e.late.skeleton", (n, *TI.LATE_SHAPES["skeleton"])).astype(np.float32),
             "inertial": filler.bellish("x.e2e.late.inertial", (n, *TI.LATE_SHAPES["inertial"]), scale=0.5).astype(np.float32)}
    labels = filler.uniform("y.e2e.late", (n,), 0, 27).astype(np.int64)
    for name, a in feats.items():
        with NumpyWriter(str(tmp_path / f"{name}_train_features.npy"), np.float32, a.shape) as w:
            for s in a:
                w.collect_next(s)
    np.save(tmp_path / "train_labels.npy", labels)
    ds = MultiModalDataset([(str(tmp_path), NumpyDatasetLoader())], "train")
    assert set(ds.features_data) == {"skeleton", "inertial"}

    model, sd = TI.late_build(gc_model="agcn")
    model = model.to(dev).train()
    hyper = dict(momentum=0.9, nesterov=True, weight_decay=1e-4)
    opt = FlatOptimizer(model.parameters(), "SGD", 0.01, **hyper)
    names = [k.replace("_model.", "") for k, _ in model.named_parameters()]
    ref_params = [sd[k] for k in names]
    ref_opt = torch.optim.SGD(ref_params, 0.01, **hyper)
    losses, ref_losses = [], []
    for x, y, idx in ClipBatches(ds, bs, shuffle=False, device=dev, resident=True):
        assert isinstance(x, dict) and x["skeleton"].is_cuda
        opt.zero_grad()
        loss = F.cross_entropy(model(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        xi = {k: torch.from_numpy(v[idx.numpy()]).double() for k, v in feats.items()}
        _, ref_loss, grads = TI.late_loss_and_grads(xi, torch.from_numpy(labels[idx.numpy()]), sd)
        ref_opt.zero_grad()
        for k, p in zip(names, ref_params):
            p.grad = grads[k] if grads[k] is not None else torch.zeros_like(p)
        ref_opt.step()
        ref_losses.append(float(ref_loss))
    assert len(losses) == 2
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 3e-4 * max(1.0, abs(b)), (losses, ref_losses)
    worst = ("", 0.0)
    for (name, p), k in zip(model.named_parameters(), names):
        want = sd[k].numpy()
        if np.abs(want).max() == 0:
            continue
        worst = max(worst, (k, rel_l2(p.detach().cpu().double().numpy(), want)), key=lambda t: t[1])
    assert worst[1] < 2e-3, worst


def test_two_ranks_of_the_real_model_through_the_self_launching_bench():
    """`python bench.py --gpus 2` with no launcher (the driver's form): bench.py starts two ranks itself; on this one-GPU box they
    share the device and exchange over gloo (FGCN_BENCH_BACKEND) -- plumbing, not a measurement -- but it is the REAL model: ONE
    16-clip batch sharded 8 + 8, HIP-graph step per rank, one flat all-reduce, and --verify-dp compares the exchanged gradient
    buffer of the replayed step with an eager step."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["FGCN_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "16", "--steps", "3", "--warmup", "1",
                        "--verify-dp", "--no-kernel-timing"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # every kernel of the step is a fixed-order libfgcn sum (data_bn and the loss included since round 3: MIOpen's BatchNorm backward
    # was the one gradient of 274 that moved under contention), so the replayed step equals the eager one bit for bit, also with
    # the two ranks sharing this GPU
    assert "verify-dp: flat gradient buffer" in r.stderr and "rel-L2 0.00e+00" in r.stderr, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["per_gpu_batch"] == 8 and out["config"]["launch"] == "hipgraph"
    assert out["other_scaling"]["scaling"] == "weak" and out["other_scaling"]["per_gpu_batch"] == 16
    assert out["value"] > 0 and abs(out["config"]["loss"]) < 20
