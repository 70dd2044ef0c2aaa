"""Data-parallel exchange on CPU: world_size-2 gloo processes.  The HIP block needs a GPU, so the replica model here
is a small stand-in nn.Module; what is under test is the N > 1 machinery bench.py uses — batch sharding, parameter
broadcast, the flat gradient buffer and its single all-reduce — against a single process that runs the same two
micro-batches sequentially (per-replica BatchNorm statistics, averaged gradients: SURVEY.md §8e)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from fusion_gcn_amd.dp import FlatGradients, broadcast_parameters, shard_batch


def make_model(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Conv1d(3, 8, 3, padding=1), nn.BatchNorm1d(8), nn.ReLU(), nn.Conv1d(8, 5, 1),
                         nn.AdaptiveAvgPool1d(1), nn.Flatten())


def data():
    g = torch.Generator().manual_seed(7)
    return torch.randn(8, 3, 16, generator=g), torch.randint(0, 5, (8,), generator=g)


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = make_model(seed=100 + rank)          # replicas start different on purpose
        broadcast_parameters(model, src=0)
        grads = FlatGradients(model.parameters())
        x, y = data()
        sl = shard_batch(x.shape[0], rank, world)
        steps = []
        for _ in range(2):
            grads.zero()
            loss = F.cross_entropy(model(x[sl]), y[sl])
            loss.backward()
            grads.all_reduce_mean()
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(grads.params, grads.views))
            steps.append(grads.flat.clone())
            with torch.no_grad():
                for p in model.parameters():
                    p -= 0.1 * p.grad
        out_q.put((rank, [s.numpy() for s in steps], float(loss)))
    finally:
        dist.destroy_process_group()


def _eager_step_worker(rank, world, port, out_q):
    """DefaultStep under a 2-rank group: its run_optimizer_step must average the gradients before the update (the base Step does
    the exchange; only GraphStep used to) and broadcast_parameters must bump the parameters' version counters."""
    from fusion_gcn_amd.session.procedures.step import DefaultStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = make_model(seed=200 + rank)
        before = [p._version for p in model.parameters()]
        broadcast_parameters(model, src=0)
        bumped = all(p._version > b for p, b in zip(model.parameters(), before))
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        step = DefaultStep()
        x, y = data()
        sl = shard_batch(x.shape[0], rank, world)
        for _ in range(2):
            opt.zero_grad()
            _, loss = step.forward(model, F.cross_entropy, x[sl], y[sl])
            step.backward(loss)
            step.run_optimizer_step(opt)
        out_q.put((rank, bumped, [p.detach().clone().numpy() for p in model.parameters()]))
    finally:
        dist.destroy_process_group()


def test_eager_steps_exchange_gradients_too():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eager_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, bumped, params = q.get(timeout=120)
        assert bumped, "broadcast_parameters must increment the version counters"
        got[rank] = params
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(got[0], got[1]):              # the replicas stayed in lock step
        assert (a == b).all()
    # and they hold what two sequential replicas with averaged gradients hold
    replicas = [make_model(seed=200) for _ in range(world)]
    x, y = data()
    for _ in range(2):
        grads = []
        for r, m in enumerate(replicas):
            m.zero_grad()
            sl = shard_batch(x.shape[0], r, world)
            F.cross_entropy(m(x[sl]), y[sl]).backward()
            grads.append([p.grad.clone() for p in m.parameters()])
        with torch.no_grad():
            for m in replicas:
                for p, ga, gb in zip(m.parameters(), *grads):
                    p -= 0.1 * (ga + gb) / world
    for mine, want in zip(got[0], replicas[0].parameters()):
        torch.testing.assert_close(torch.from_numpy(mine), want.detach(), rtol=1e-5, atol=1e-6)


class _PartlyUsed(nn.Module):
    """`a` is used by every rank, `b` by rank 1 only, `c` by no rank."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.a, self.b, self.c = nn.Linear(4, 3), nn.Linear(4, 3), nn.Linear(4, 3)

    def forward(self, x, use_b):
        return self.a(x) + (self.b(x) if use_b else 0)


def _unused_worker(rank, world, port, out_q):
    from fusion_gcn_amd.session.procedures.step import DefaultStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _PartlyUsed()
        opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=0.5)
        step = DefaultStep()
        g = torch.Generator().manual_seed(3 + rank)
        for _ in range(2):
            opt.zero_grad()
            x, y = torch.randn(6, 4, generator=g), torch.randint(0, 3, (6,), generator=g)
            loss = F.cross_entropy(model(x, use_b=rank == 1), y)
            step.backward(loss)
            step.run_optimizer_step(opt)
        out_q.put((rank, {k: v.detach().clone().numpy() for k, v in model.state_dict().items()},
                   [p.grad is None for p in model.c.parameters()], [p.grad is None for p in model.b.parameters()]))
    finally:
        dist.destroy_process_group()


def test_unused_parameters_keep_the_single_rank_semantics():
    """A parameter that received no gradient on ANY rank must come back from the exchange with p.grad = None, so that torch.optim
    skips it (no weight decay, no momentum) as in the single-rank run (the reference's behaviour); one that some rank
    used travels as zeros from the others and is updated identically everywhere."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unused_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, sd, c_none, b_none = q.get(timeout=120)
        got[rank] = sd
        assert all(c_none) and not any(b_none), (rank, c_none, b_none)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fresh = _PartlyUsed().state_dict()
    for k in fresh:
        assert (got[0][k] == got[1][k]).all(), k                      # replicas in lock step
        if k.startswith("c."):
            assert (got[0][k] == fresh[k].numpy()).all(), k           # untouched: no decay was applied
        else:
            assert not (got[0][k] == fresh[k].numpy()).all(), k


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_replicas_match_sequential_micro_batches():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, steps, loss = q.get(timeout=120)
        results[rank] = steps
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # both ranks hold identical averaged gradients
    for a, b in zip(results[0], results[1]):
        assert (a == b).all()

    # single-process emulation: rank-0 weights, the two shards run one after the other, gradients averaged
    replicas = [make_model(seed=100) for _ in range(world)]
    x, y = data()
    for step in range(2):
        flats = []
        for r, m in enumerate(replicas):
            for p in m.parameters():
                p.grad = None
            sl = shard_batch(x.shape[0], r, world)
            F.cross_entropy(m(x[sl]), y[sl]).backward()
            fg = FlatGradients(m.parameters())
            fg.gather()
            flats.append(fg.flat.clone())
        mean = sum(flats) / world
        torch.testing.assert_close(torch.from_numpy(results[0][step]), mean, rtol=1e-5, atol=1e-6)
        with torch.no_grad():
            for m in replicas:
                fg = FlatGradients(m.parameters())
                for p, v in zip(fg.params, [mean[o:o + p.numel()].view_as(p) for p, o in
                                            zip(fg.params, _offsets(fg.params))]):
                    p -= 0.1 * v


def _offsets(params):
    out, total = [], 0
    for p in params:
        out.append(total)
        total += (p.numel() + 3) // 4 * 4
    return out


def test_shard_batch_contract():
    assert [shard_batch(64, r, 8) for r in (0, 7)] == [slice(0, 8), slice(56, 64)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def test_flat_gradients_single_process():
    m = make_model(seed=3)
    fg = FlatGradients(m.parameters())
    x, y = data()
    fg.zero()
    assert all(p.grad is None for p in m.parameters())
    F.cross_entropy(m(x), y).backward()
    want = [p.grad.clone() for p in m.parameters()]
    fg.all_reduce_mean()      # no process group: gather only
    for p, w in zip(m.parameters(), want):
        torch.testing.assert_close(p.grad, w)
        assert p.grad.data_ptr() >= fg.flat.data_ptr()
    assert fg.flat.numel() % 4 == 0
