#!/usr/bin/env python3
"""Step time of the MS-G3D model at the size of config/utd-mhad/skeleton/msg3d.yaml (model: msg3d, batch 8, UTD-MHAD skeleton
clips: M = 1, T = 128 (skeleton_max_sequence_length), V = 20, C = 3, 27 classes): train-mode fwd+bwd on one MI355X in the math
modes, next to the float32 CPU oracle (the stock-torch restatement of the reference model) on the host cores."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.msg3d.msg3d import Model
    from fusion_gcn_amd.util import Graph
    dev = torch.device("cuda:0")
    shape, classes = (1, args.frames, 20, 3), 27
    g = Graph(utd.skeleton_edges, center_joint=utd.center_joint)
    torch.manual_seed(1)
    model = Model({"skeleton": shape}, classes, g).to(dev).train()
    x = torch.randn(args.batch, *shape, device=dev)
    y = torch.randint(0, classes, (args.batch,), device=dev)
    out = {"workload": f"MS-G3D fwd+bwd, batch {args.batch}, (M,T,V,C)={shape}, {classes} classes",
           "parameters": sum(p.numel() for p in model.parameters())}
    for mode in ("f32", "bf16x3", "bf16"):
        with ops.math_mode(mode):
            for _ in range(2):
                model.zero_grad(set_to_none=True)
                F.cross_entropy(model(x), y).backward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model.zero_grad(set_to_none=True)
                loss = F.cross_entropy(model(x), y)
                loss.backward()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
        out[mode] = {"ms_per_step": round(1e3 * dt, 2), "clips_per_s": round(args.batch / dt, 1), "loss": round(float(loss), 5)}
    if args.cpu:
        from oracle import msg3d_oracle as O
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        a = g.get_adjacency_matrix().astype("float64")
        xc, yc = x.cpu(), y.cpu()
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        O.loss_and_grads(xc[:2], yc[:2], sd, a)
        t0 = time.perf_counter()
        O.loss_and_grads(xc, yc, sd, a)
        dt = time.perf_counter() - t0
        out["cpu_oracle"] = {"ms_per_step": round(1e3 * dt, 1), "clips_per_s": round(args.batch / dt, 2), "threads": torch.get_num_threads()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
