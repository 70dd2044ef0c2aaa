#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generate tests/golden/msg3d.npz (+ the key manifest inside it) by IMPORTING THE REFERENCE's MS-G3D model
(torch_src/models/msg3d/msg3d.py:113-182; build container only; import recipe of oracle/gen_golden.py).

    python oracle/gen_golden_msg3d.py

The reference model is constructed, its state dict filled by oracle.filler, inputs come from the same filler; only OUTPUTS are
stored (float64): eval / train logits, loss, per-parameter gradient norms, small gradients in full, running statistics of a few
BatchNorms after the training forward, the adjacency stacks the model builds, the state-dict key list in order, and -- for the
model built under torch.manual_seed(1) -- the initial-state fingerprints."""
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

from util.graph import Graph  # noqa: E402  (reference)
import models.msg3d.msg3d as ref_msg3d  # noqa: E402  (reference)
from datasets.utd_mhad import constants as utd  # noqa: E402  (reference)
from datasets.ntu_rgb_d import constants as ntu  # noqa: E402  (reference)

from oracle import filler  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "msg3d.npz")
torch.set_num_threads(8)


def main():
    store = {}
    for tag, shape, classes, consts in (("utd", (2, 1, 16, 20, 3), 27, utd), ("ntu", (2, 2, 12, 25, 3), 60, ntu)):
        g = Graph(consts.skeleton_edges, center_joint=consts.center_joint)
        n, m, t, v, c = shape
        torch.manual_seed(1)
        model = ref_msg3d.Model({"skeleton": (m, t, v, c)}, classes, g)
        sd0 = model.state_dict()
        store[f"{tag}.keys"] = np.array(list(sd0.keys()))
        store[f"{tag}.init_fingerprint"] = np.array([[float(p.double().sum()), float((p.double() ** 2).sum())] for p in sd0.values()])
        store[f"{tag}.a_binary"] = g.get_adjacency_matrix().astype(np.float64)
        store[f"{tag}.A_powers.sgcn1"] = model.sgcn1[0].A_powers.numpy().astype(np.float64)
        store[f"{tag}.A_scales.w3"] = model.gcn3d1.gcn3d[0].gcn3d[1].A_scales.numpy().astype(np.float64)
        store[f"{tag}.A_scales.w5"] = model.gcn3d1.gcn3d[1].gcn3d[1].A_scales.numpy().astype(np.float64)
        model = model.double()
        model.sgcn1[0].A_powers = model.sgcn1[0].A_powers.double()
        filler.fill_state_dict(model.state_dict())
        x = torch.from_numpy(filler.skeleton_input(f"x.msg3d.{tag}", shape, empty_second_body=(m > 1))).double()
        labels = torch.from_numpy(filler.uniform(f"y.msg3d.{tag}", (n,), 0, classes).astype(np.int64))
        store[f"{tag}.labels"] = labels.numpy()
        before = {k: v_.clone() for k, v_ in model.state_dict().items()}
        model.eval()
        store[f"{tag}.eval.logits"] = model(x).detach().numpy()
        model.train()
        logits = model(x)
        loss = F.cross_entropy(logits, labels)
        loss.backward()
        store[f"{tag}.train.logits"] = logits.detach().numpy()
        store[f"{tag}.train.loss"] = loss.detach().numpy()
        for name, p in model.named_parameters():
            store[f"{tag}.gl2.{name}"] = p.grad.norm().numpy()
            if p.numel() <= 2048:
                store[f"{tag}.grad.{name}"] = p.grad.numpy()
        for k, v_ in model.state_dict().items():
            if k.endswith(("running_mean", "running_var")) and k.split(".")[0] in ("data_bn", "tcn1", "gcn3d2", "sgcn3"):
                store[f"{tag}.after.{k}"] = v_.numpy().copy()
        model.load_state_dict(before)
    store["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(OUT, **store)
    print(OUT, os.path.getsize(OUT), "bytes,", len(store), "entries")


if __name__ == "__main__":
    main()
