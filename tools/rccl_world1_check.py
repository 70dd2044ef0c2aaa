#!/usr/bin/env python3
"""What a 1-GPU box can prove about the data-parallel exchange (SURVEY.md section 8 row e; the reference has no distributed
code): librccl loads, an `nccl` (= RCCL) communicator binds to the device, and the stream ordering HIP-graph replay ->
collective -> next replay is right.  One rank, world size 1 (never two ranks on one device with `nccl`).

Run as its own process (tests/test_rccl_gpu.py starts it as a child): the real 60-class NTU model and its 13.9 MB flat gradient
buffer, `GraphStep` replays + `FlatGradients.all_reduce_mean(always=True)` for --steps steps.  Checks, every step: the world-1
all-reduce (SUM over one rank) leaves the buffer bit-for-bit unchanged, and the loss equals the step run before any process
group existed.  Prints one JSON line with the per-step cost of the collective behind a graph replay."""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def _rccl_version() -> str:
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001 - informational field only
        return f"unknown ({type(e).__name__})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--frames", type=int, default=300)
    args = ap.parse_args()
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from fusion_gcn_amd.dp import FlatGradients, broadcast_parameters
    from fusion_gcn_amd.loss import CrossEntropyLoss
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.session.procedures import GraphStep
    from fusion_gcn_amd.util import Graph
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = Model((2, args.frames, 25, 3), 60, Graph(ntu.skeleton_edges, center_joint=ntu.center_joint))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gcn1.bn.weight"):
                p.fill_(1.0)
    model = model.to(dev).train()
    loss_function = CrossEntropyLoss()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(args.clips, 2, args.frames, 25, 3, generator=g).to(dev)
    y = torch.randint(0, 60, (args.clips,), generator=g).to(dev)

    # the step with no process group anywhere (eager launches)
    loss0 = loss_function(model(x), y)
    loss0.backward()
    torch.cuda.synchronize()
    want_loss = float(loss0)
    want_flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    del loss0
    model.zero_grad(set_to_none=True)

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    init_s = time.perf_counter() - t0
    broadcast_parameters(model)                              # world 1: returns at once
    grads = FlatGradients(model.parameters())
    step = GraphStep(grads=grads, data_parallel=False)       # the exchange is issued here, explicitly, for a single rank too
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    cost_us, losses, unchanged = [], [], []
    for i in range(args.steps):
        model.zero_grad(set_to_none=True)
        _, loss = step.forward(model, loss_function, x, y)     # record on the first call, replay afterwards
        step.backward(loss)
        before = grads.flat.clone()
        ev[0].record()
        grads.all_reduce_mean(always=True)                       # ONE RCCL all-reduce of the flat buffer on the current stream
        ev[1].record()
        torch.cuda.synchronize()
        cost_us.append(1e3 * ev[0].elapsed_time(ev[1]))
        unchanged.append(bool(torch.equal(before, grads.flat)))
        losses.append(float(loss))
    got_flat = torch.cat([v.reshape(-1) for v in grads.views])
    rel = float((got_flat - want_flat).norm() / want_flat.norm())
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "init_s": round(init_s, 3), "steps": args.steps,
           "replays": step.replays, "flat_bytes": grads.flat.numel() * 4, "buffer_unchanged_by_the_collective": all(unchanged),
           "loss": losses[-1], "loss_without_process_group": want_loss,
           "loss_equal": all(abs(v - want_loss) <= 1e-6 * abs(want_loss) for v in losses),
           "flat_grad_rel_l2_vs_eager_without_group": rel,
           "flat_grad_bitwise_equal_to_eager_without_group": bool(torch.equal(got_flat, want_flat)),
           "all_reduce_us_first": round(cost_us[0], 1), "all_reduce_us_after_replay_mean": round(sum(cost_us[2:]) / max(1, len(cost_us) - 2), 1),
           "rccl_version": _rccl_version()}
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)
    ok = out["buffer_unchanged_by_the_collective"] and out["loss_equal"] and rel < 1e-5 and out["backend"] == "nccl"
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
