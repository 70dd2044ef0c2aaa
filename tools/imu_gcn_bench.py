#!/usr/bin/env python3
"""Step time of the IMU graph model of config/utd-mhad/imu/imu_gcn_v1_stgcn.yaml (mode imu_gcn, gc_model stgcn, node_per_value,
inner_feature_dim 512, 10 layers, batch 8; UTD-MHAD inertial sequences resampled to 326 x 6 -> 1956 nodes): fwd+bwd on one
MI355X in the math modes, with the algorithmic FLOPs, next to the float32 CPU oracle on a 2-sample slice."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=326)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu", action="store_true", help="also time the oracle on the host cores (2 samples)")
    ap.add_argument("--graph", action="store_true", help="also time the step replayed from a recorded HIP graph (GraphStep)")
    ap.add_argument("--late", action="store_true",
                    help="instead: mode skeleton_imu_gcn_late_fusion as config/utd-mhad/skeleton+imu/late_fusion/*.yaml (skeleton "
                         "(1, 128, 20, 3) + inertial (326, 6), gc_model agcn, node_per_sensor, num_signals 2, batch 8)")
    args = ap.parse_args()
    if args.late:
        return late(args)
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.models.mmargcn.mmargcn import Model
    dev = torch.device("cuda:0")
    shape, classes = (args.frames, 6), 27
    torch.manual_seed(1)
    model = Model({"inertial": shape}, classes, None, mode="imu_gcn", gc_model="stgcn", graph_node_format="node_per_value",
                  inner_feature_dim=args.width, num_layers=args.layers).to(dev).train()
    x = torch.randn(args.batch, *shape, device=dev)
    y = torch.randint(0, classes, (args.batch,), device=dev)
    V = shape[0] * shape[1]
    flops, f = 0.0, 1
    for layer in model._model.gcn.layers:
        o = layer.out_features
        flops += 2.0 * args.batch * V * f * o * (2 if layer.res_kind == "conv" else 1) + 2.0 * args.batch * o * V * V
        f = o
    flops *= 3
    out = {"nodes": V, "batch": args.batch, "algorithmic_gflop_per_step": round(flops / 1e9, 1)}
    from steptime import time_step
    # dense matrix peaks (MI355X_MICROARCH.md) per arithmetic: exact f32 MFMA, bf16 / 6 products, f16 / 3 products, bf16
    peaks = {"f32": 157.3, "bf16x3": 2500.0 / 6, "f16x2": 2500.0 / 3, "bf16": 2500.0}
    for mode in ("f32", "bf16x3", "f16x2", "bf16"):
        with ops.math_mode(mode):
            t = time_step(model, x, y, args.steps, graph=args.graph)
        best = min(t["eager"]["ms_per_step"], t["graph"]["ms_per_step"]) if args.graph else t["eager"]["ms_per_step"]
        out[mode] = {"ms_per_step": t["eager"]["ms_per_step"], "samples_per_s": t["eager"]["per_s"],
                     "tflops": round(flops / t["eager"]["ms_per_step"] / 1e9, 1), "loss": t["eager"]["loss"],
                     # the whole step against the matrix roof of its arithmetic (the step is matrix-bound: 1956 x 1956 adjacency products
                     # and 512..4096-wide feature GEMMs; algorithmic FLOPs = 3 x forward)
                     "roofline": {"bound": "mfma", "achieved": round(flops / best / 1e9, 1), "peak": round(peaks[mode], 1),
                                  "unit": "TFLOP/s", "frac": round(flops / best / 1e9 / peaks[mode], 3), "what": "whole fwd+bwd step"}}
        if args.graph:
            out[mode]["graph"] = dict(t["graph"], tflops=round(flops / t["graph"]["ms_per_step"] / 1e9, 1))
    if args.cpu:
        from oracle import imu_gcn_oracle as O
        sd = {k.replace("_model.", ""): v.detach().cpu() for k, v in model.state_dict().items()}
        xs, ys = x[:2].cpu(), y[:2].cpu()
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        O.loss_and_grads(xs, ys, sd)
        t0 = time.perf_counter()
        O.loss_and_grads(xs, ys, sd)
        dt = time.perf_counter() - t0
        out["cpu_oracle"] = {"samples_per_s": round(2 / dt, 3), "threads": torch.get_num_threads(), "sample": "2 samples, 1 iteration"}
    print(json.dumps(out))


def late(args):
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.mmargcn.mmargcn import Model
    from fusion_gcn_amd.util import Graph
    dev = torch.device("cuda:0")
    shapes, classes = {"skeleton": (1, 128, 20, 3), "inertial": (args.frames, 6)}, 27
    torch.manual_seed(1)
    model = Model(shapes, classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint), mode="skeleton_imu_gcn_late_fusion",
                  graph_node_format="node_per_sensor", num_signals=2, gc_model="agcn", fusion="concatenate").to(dev).train()
    x = {k: torch.randn(args.batch, *v, device=dev) for k, v in shapes.items()}
    y = torch.randint(0, classes, (args.batch,), device=dev)
    out = {"mode": "skeleton_imu_gcn_late_fusion", "imu_nodes": args.frames * 2, "batch": args.batch}
    # algorithmic FLOPs of fwd + bwd (= 3 x forward), counted from the module tree: every AGCN block of the skeleton branch by SURVEY.md
    # section 8d's formula (M = 1, T = 128, V = 20), every AGCNGraphConvolution of the IMU branch (graph_convolution.py:56-113: theta | phi
    # embeddings, V x V scores and aggregation per subset, conv_d, down) -- the classifier and the poolings are < 0.1 %
    from fusion_gcn_amd.models.mmargcn.agcn import SpatialTemporalConv
    from fusion_gcn_amd.models.mmargcn.graph_convolution import AGCNGraphConvolution
    B = args.batch
    flops = 0.0
    T, V = shapes["skeleton"][1], shapes["skeleton"][2]
    for m in model.modules():
        if isinstance(m, SpatialTemporalConv):
            c = m.cfg
            ic, Tp = c.cout // 4, (T - 1) // c.stride + 1
            f = B * T * V * (12 * c.cin * ic + 6 * V * c.cin + 6 * c.cin * c.cout + (2 * c.cin * c.cout if c.has_down else 0))
            f += 6 * V * V * ic * T * B + B * Tp * V * (18 * c.cout * c.cout + (2 * c.cin * c.cout if c.residual == "conv" else 0))
            flops += 3 * f
            T = Tp
        elif isinstance(m, AGCNGraphConvolution):
            n, f_in, f_out, ic, K = args.frames * 2, m.in_features, m.out_features, m.inter_c, m.num_subset
            f = B * K * (4.0 * n * f_in * ic + 2.0 * n * n * ic + 2.0 * n * n * f_in + 2.0 * n * f_in * f_out)
            f += 2.0 * B * n * f_in * f_out if m.has_down else 0.0
            flops += 3 * f
    out["algorithmic_gflop_per_step"] = round(flops / 1e9, 1)
    peaks = {"f32": 157.3, "bf16x3": 2500.0 / 6}
    from steptime import time_step
    for mode in ("f32", "bf16x3"):
        with ops.math_mode(mode):
            t = time_step(model, x, y, args.steps, graph=args.graph)
        best = min(t["eager"]["ms_per_step"], t["graph"]["ms_per_step"]) if args.graph else t["eager"]["ms_per_step"]
        out[mode] = {"ms_per_step": t["eager"]["ms_per_step"], "samples_per_s": t["eager"]["per_s"], "loss": t["eager"]["loss"],
                     # the whole step against the matrix roof of its arithmetic; in bf16x3 the per-sample V x V products of the IMU branch
                     # still run on exact-f32 row GEMMs (fgcn_rows_gemm_batched), so the bf16x3 roof overstates what this step can reach
                     "roofline": {"bound": "mfma", "achieved": round(flops / best / 1e9, 1), "peak": round(peaks[mode], 1),
                                  "unit": "TFLOP/s", "frac": round(flops / best / 1e9 / peaks[mode], 3), "what": "whole fwd+bwd step"}}
        if args.graph:
            out[mode]["graph"] = t["graph"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
