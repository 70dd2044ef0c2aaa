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
    ap.add_argument("--modes", default="f32,bf16x3,bf16")
    ap.add_argument("--graph", action="store_true", help="also time the step replayed from one captured HIP graph")
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
    params = list(model.parameters())
    # everything runs on one non-default stream: autograd pins each parameter's gradient accumulation to the stream of its first
    # use, and a capture that has to synchronise with the legacy default stream is invalid (HIP ends it with a crash, not an error)
    work = torch.cuda.Stream()
    work.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(work)
    graphs = []                                   # kept alive to the end: a graph's pool is shared with the tensors it produced
    def step() -> torch.Tensor:
        for p in params:
            p.grad = None
        loss = F.cross_entropy(model(x), y)
        loss.backward()
        return loss                                   # the caller must not hold it across a capture (see --graph below)

    # algorithmic FLOPs of one forward pass, counted at the C-ABI boundary: every GEMM-shaped launch of a no-grad forward (row GEMMs
    # incl. the dilated temporal convolutions and the stacked-adjacency products, the halo conv, the persistent 1x1 GEMM); fwd+bwd = 3x
    counted = [0.0]

    def counting(name, flop_of):
        real = getattr(ops, name)

        def wrapper(*a, **kw):
            counted[0] += flop_of(*a, **kw)
            return real(*a, **kw)
        setattr(ops, name, wrapper)
        return real
    rows_of = lambda t: t.shape[0] * t.shape[1] * t.shape[2]                       # noqa: E731
    saved = {"rows_gemm": counting("rows_gemm", lambda x, w, out, *, K, N, **kw: 2.0 * rows_of(out) * (kw["tmap"][0] if kw.get("tmap") else 1) * K * N),
             "pw_gemm": counting("pw_gemm", lambda x, w3, out, **kw: 2.0 * rows_of(out) * x.shape[3] * out.shape[3]),
             "tconv_halo": counting("tconv_halo", lambda x, w, out, *, taps, **kw: 2.0 * rows_of(out) * taps * x.shape[3] * out.shape[3])}
    with torch.no_grad():
        model(x)
    for k, v in saved.items():
        setattr(ops, k, v)
    flops = 3.0 * counted[0]
    out["algorithmic_gflop_per_step"] = round(flops / 1e9, 1)
    peaks = {"f32": 157.3, "bf16x3": 2500.0 / 6, "f16x2": 2500.0 / 3, "bf16": 2500.0}     # dense matrix peak of each arithmetic, TFLOP/s

    def roofline(ms):
        return {"bound": "mfma", "achieved": round(flops / ms / 1e9, 1), "peak": round(peaks[mode], 1), "unit": "TFLOP/s",
                "frac": round(flops / ms / 1e9 / peaks[mode], 3), "what": "whole fwd+bwd step (3 x the forward's GEMM FLOPs)"}

    for mode in args.modes.split(","):
        with ops.math_mode(mode):
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            eager_loss = float(loss.detach())
            del loss
            out[mode] = {"ms_per_step": round(1e3 * dt, 2), "clips_per_s": round(args.batch / dt, 1), "loss": round(eager_loss, 5),
                         "roofline": roofline(1e3 * dt)}
            if args.graph:
                # Every lazily built buffer exists after the eager steps: record forward + backward once and replay.  No reference
                # to an earlier step's autograd graph may be alive here (tools/probes/msg3d_graph_probe.py: it pins the
                # parameters' gradient accumulators to the other stream and later captures replay garbage), and the replayed
                # loss is checked against the eager one.
                for p in params:
                    p.grad = None
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    gloss = F.cross_entropy(model(x), y)
                    gloss.backward()
                graphs.append((graph, gloss))
                for _ in range(3):
                    graph.replay()
                    torch.cuda.synchronize()
                    got = float(gloss.detach())
                    if got != eager_loss:
                        raise RuntimeError(f"{mode}: graph replay loss {got} != eager loss {eager_loss}")
                t0 = time.perf_counter()
                for _ in range(args.steps * 4):
                    graph.replay()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / (args.steps * 4)
                got = float(gloss.detach())
                if got != eager_loss:
                    raise RuntimeError(f"{mode}: graph replay loss {got} != eager loss {eager_loss} after the timed replays")
                out[mode]["graph"] = {"ms_per_step": round(1e3 * dt, 2), "clips_per_s": round(args.batch / dt, 1), "loss": round(got, 5),
                                      "roofline": roofline(1e3 * dt)}
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
