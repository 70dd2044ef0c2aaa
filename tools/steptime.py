"""Step timing shared by the tools/*_bench.py scripts: forward + loss + backward of one resident batch, launched eagerly and
(``graph=True``) replayed from the HIP graph that fusion_gcn_amd.session.procedures.GraphStep records (which also checks the replay
against the eager step before it is used)."""
import time

import torch
import torch.nn.functional as F


def time_step(model, features, label, steps: int, graph: bool = False, warmup: int = 2) -> dict:
    from fusion_gcn_amd.session.procedures import DefaultStep, GraphStep
    params = [p for p in model.parameters() if p.requires_grad]
    out = {}
    for name, step in (("eager", DefaultStep()),) + ((("graph", GraphStep()),) if graph else ()):
        def one():
            for p in params:
                p.grad = None                       # optimizer.zero_grad()
            _, loss = step.forward(model, F.cross_entropy, features, label)
            step.backward(loss)
            return loss.detach()
        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        n = steps * (4 if name == "graph" else 1)
        t0 = time.perf_counter()
        for _ in range(n):
            loss = one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[name] = {"ms_per_step": round(1e3 * dt, 2), "per_s": round(label.shape[0] / dt, 1), "loss": round(float(loss), 5)}
        del loss
        for p in params:
            p.grad = None
    return out
