"""Inference throughput of the AGCN model at the headline shape: eval mode, torch.no_grad(), 64 clips, HIP-graph replay;
the fused output stages (paths.fused_inference) against the two-pass eval path.  python tools/infer_bench.py [--batch 64] [--math bf16x3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from fusion_gcn_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--math", default="bf16x3")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.set_math_mode(args.math)
    model = bench.build_model(dev).eval()
    x = torch.randn(args.batch, 2, 300, 25, 3, device=dev)
    out = {"workload": f"AGCN 10-block inference (eval-mode BatchNorm, no autograd), (N,C,T,V,M)=({args.batch},3,300,25,2)", "math": args.math}
    for fused in (False, True):
        ops.paths().fused_inference = fused
        with torch.no_grad():
            for _ in range(3):
                y = model(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = model(x)
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
        out["fused_output_stages" if fused else "two_pass_eval"] = {"ms_per_batch": round(1e3 * dt, 3), "clips_per_s": round(args.batch / dt, 1),
                                                                    "logit_checksum": float(y.double().sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
