"""GraphStep (fusion_gcn_amd/session/procedures/step.py): training through recorded HIP graphs must be the eager training --
same losses, same parameters, same BatchNorm statistics -- over several batch shapes per process (a ragged last batch = a second
recording), with returned losses held by the caller (the case tools/probes/msg3d_graph_probe.py found to invalidate later
recordings when streams are mixed), with gradient accumulation, and for the three model families."""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import filler

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def agcn(shape=(1, 24, 20, 3), classes=27):
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.util import Graph
    model = Model(shape, classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint))
    filler.fill_state_dict(model.state_dict())
    return model


def msg3d(shape=(1, 16, 20, 3), classes=27):
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.msg3d.msg3d import Model
    from fusion_gcn_amd.util import Graph
    torch.manual_seed(3)
    return Model({"skeleton": shape}, classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint))


def batches(sizes, shape, classes, seed=5):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(n, *shape, generator=g).to(DEV), torch.randint(0, classes, (n,), generator=g).to(DEV), torch.arange(n)) for n in sizes]


def train(model, processor, data, lr=0.01):
    from fusion_gcn_amd.optim import FlatOptimizer
    from fusion_gcn_amd.session.session import Session
    model = model.to(DEV).train()
    opt = FlatOptimizer(model.parameters(), "SGD", lr, momentum=0.9)

    class Keep:                       # holds every loss / prediction tensor it is handed, like a metrics container would
        def __init__(self):
            self.losses, self.preds = [], []

        def update_training(self, loss, pair, m, idx):
            self.losses.append(loss)
            self.preds.append(pair[0])

        def format_training(self):
            return ""
    keep = Keep()
    Session.train_epoch(processor, model, F.cross_entropy, data, opt, None, keep)
    torch.cuda.synchronize()
    return model, keep


def assert_same_training(a, b, keep_a, keep_b, tol=1e-6):
    for la, lb in zip(keep_a.losses, keep_b.losses):
        assert abs(float(la) - float(lb)) <= tol * max(1.0, abs(float(lb))), ([float(t) for t in keep_a.losses], [float(t) for t in keep_b.losses])
    for pa, pb in zip(keep_a.preds, keep_b.preds):
        assert float((pa - pb).abs().max()) <= tol * max(1.0, float(pb.abs().max()))
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb)
    for k in sa:
        va, vb = sa[k], sb[k]
        if va.is_floating_point():
            assert float((va - vb).abs().max()) <= tol * max(1e-3, float(vb.abs().max())), k
        else:
            assert torch.equal(va, vb), k                # BatchNorm batch counters: the warm-up and verification runs rolled back


def test_graph_training_is_the_eager_training_over_two_batch_shapes(fgcn_math):
    from fusion_gcn_amd.session.procedures import DefaultBatchProcessor, DefaultStep, GraphStep
    shape, classes = (1, 24, 20, 3), 27
    data = batches([4, 4, 4, 3, 4, 3], shape, classes)           # the 3-clip batches are a second recording
    base = agcn(shape, classes)
    step = GraphStep()
    eager, keep_e = train(copy.deepcopy(base), DefaultBatchProcessor(DefaultStep()), data)
    graph, keep_g = train(copy.deepcopy(base), DefaultBatchProcessor(step), data)
    assert len(step._recorded) == 2 and step.replays == 6
    assert_same_training(graph, eager, keep_g, keep_e)
    # evaluation goes through the eager forward of the same step object; the weights the replays trained are the model's own
    from fusion_gcn_amd.session.session import Session
    seen = []

    class Val:
        def update_validation(self, loss, pair, m, idx):
            seen.append(float(loss))

        def format_all(self):
            return ""
    Session.validate_epoch(DefaultBatchProcessor(step), graph, F.cross_entropy, data[:2], None, Val())
    Session.validate_epoch(DefaultBatchProcessor(DefaultStep()), eager, F.cross_entropy, data[:2], None, Val())
    assert seen[:2] == pytest.approx(seen[2:], rel=1e-5) and step.replays == 6


def test_graph_step_accumulates_micro_batches(fgcn_math):
    from fusion_gcn_amd.session.procedures import DefaultStep, GradientAccumulationBatchProcessor, GraphStep
    shape, classes = (1, 24, 20, 3), 27
    data = batches([6, 6], shape, classes)
    base = agcn(shape, classes)
    step = GraphStep()
    eager, keep_e = train(copy.deepcopy(base), GradientAccumulationBatchProcessor(DefaultStep(), 6, 2), data)
    graph, keep_g = train(copy.deepcopy(base), GradientAccumulationBatchProcessor(step, 6, 2), data)
    assert len(step._recorded) == 1 and step.replays == 6 and len(keep_g.losses) == 6
    assert_same_training(graph, eager, keep_g, keep_e)


def test_graph_step_msg3d_and_late_fusion(fgcn_math):
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.mmargcn.mmargcn import Model as MM
    from fusion_gcn_amd.session.procedures import DefaultBatchProcessor, DefaultStep, GraphStep
    from fusion_gcn_amd.util import Graph
    shape, classes = (1, 16, 20, 3), 27
    data = batches([2, 2, 3], shape, classes)
    base = msg3d(shape, classes)
    eager, keep_e = train(copy.deepcopy(base), DefaultBatchProcessor(DefaultStep()), data)
    graph, keep_g = train(copy.deepcopy(base), DefaultBatchProcessor(GraphStep()), data)
    assert_same_training(graph, eager, keep_g, keep_e)
    # dictionary features: skeleton + inertial, late fusion with the AGCN IMU branch
    shapes = {"skeleton": (1, 16, 20, 3), "inertial": (12, 6)}
    torch.manual_seed(2)
    base = MM(shapes, classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint), mode="skeleton_imu_gcn_late_fusion",
              graph_node_format="node_per_sensor", num_signals=2, gc_model="agcn", fusion="concatenate")
    g = torch.Generator().manual_seed(9)
    data = [({k: torch.randn(n, *s, generator=g).to(DEV) for k, s in shapes.items()}, torch.randint(0, classes, (n,), generator=g).to(DEV),
             torch.arange(n)) for n in (3, 3, 2)]
    eager, keep_e = train(copy.deepcopy(base), DefaultBatchProcessor(DefaultStep()), data)
    graph, keep_g = train(copy.deepcopy(base), DefaultBatchProcessor(GraphStep()), data)
    assert_same_training(graph, eager, keep_g, keep_e)


def test_graph_step_rejects_foreign_gradients_and_records_again_after_a_move():
    from fusion_gcn_amd.session.procedures import GraphStep
    shape, classes = (1, 24, 20, 3), 27
    (x, y, _), = batches([2], shape, classes)
    model = agcn(shape, classes).to(DEV).train()
    F.cross_entropy(model(x), y).backward()                          # an eager step first (nothing of it is kept), then the switch
    model.zero_grad()
    step = GraphStep()
    y_pred, loss = step.forward(model, F.cross_entropy, x, y)
    step.backward(loss)
    first = [p.grad.clone() for p in model.parameters()]
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(step.grads.params, step.grads.views))
    y_pred, loss = step.forward(model, F.cross_entropy, x, y)        # no zero_grad in between: the gradients accumulate
    torch.cuda.synchronize()
    for p, g in zip(model.parameters(), first):
        assert torch.allclose(p.grad, 2 * g, rtol=1e-6, atol=1e-12)
    next(model.parameters()).grad = torch.zeros_like(next(model.parameters()))
    with pytest.raises(RuntimeError, match="neither None nor views"):
        step.forward(model, F.cross_entropy, x, y)
    # re-homed parameters (a FlatOptimizer created afterwards): the recording reads the old addresses and must be redone
    from fusion_gcn_amd.optim import FlatOptimizer
    opt = FlatOptimizer(model.parameters(), "SGD", 0.01, grads=step.grads)
    opt.zero_grad()
    step.forward(model, F.cross_entropy, x, y)
    assert step.replays == 3 and len(step._recorded) == 1
    torch.cuda.synchronize()
    for p, g in zip(model.parameters(), first):
        assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-12)
    # at most max_shapes recordings are kept: the oldest goes, and is recorded again when its shape returns
    small = GraphStep(grads=step.grads, max_shapes=1)
    for n in (2, 1, 2):
        opt.zero_grad()
        small.forward(model, F.cross_entropy, x[:n].contiguous(), y[:n].contiguous())
        assert len(small._recorded) == 1
    torch.cuda.synchronize()
    for p, g in zip(model.parameters(), first):
        assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-12)


def test_recordings_of_two_math_modes_keep_their_packed_weights():
    """A recording reads the blocks' packed weight buffers and the re-pack tables through raw pointers.  A forward in another
    math mode replaces every block's packed set and the model's re-pack plan (and broadcast_parameters' / drop_packed drops
    them): the earlier recording must keep its own alive (GraphStep pins them) and still replay the eager step -- with the
    CURRENT parameter values, since its recorded re-pack launch reads the parameters' homes."""
    import gc
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.session.procedures import GraphStep
    shape, classes = (1, 24, 20, 3), 27
    (x, y, _), = batches([3], shape, classes)
    model = agcn(shape, classes).to(DEV).train()
    step = GraphStep()

    def eager(mode):
        model.zero_grad()
        with ops.math_mode(mode):
            loss = F.cross_entropy(model(x), y)
            loss.backward()
        out = float(loss), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
        model.zero_grad()
        return out

    def replay(mode):
        with ops.math_mode(mode):
            _, loss = step.forward(model, F.cross_entropy, x, y)
        torch.cuda.synchronize()
        model.zero_grad()                                 # p.grad = None; the flat buffer keeps the step's gradients until the next step
        return float(loss), None

    replay("bf16x3")                                     # recording 1
    pinned = [id(o) for rec in step._recorded.values() for o in rec.pins]
    assert pinned, "the recording must hold the packed sets it reads"
    replay("f32")                                        # recording 2: every block's packed set and the plan are replaced
    assert len(step._recorded) == 2
    for blk in model.modules():
        if hasattr(blk, "drop_packed"):
            blk.drop_packed()                            # what a parameter broadcast does
    model._pack_plan = None
    gc.collect()
    junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(64)]     # reuse whatever memory was freed
    del junk
    with torch.no_grad():                                # an "optimizer update" behind the recordings
        for p in model.parameters():
            p.mul_(1.01)
    for mode in ("bf16x3", "f32", "bf16x3"):
        want_loss, want = eager(mode)
        got_loss, got = replay(mode)
        assert abs(got_loss - want_loss) <= 1e-6 * abs(want_loss), (mode, got_loss, want_loss)
        views = torch.cat([v.reshape(-1) for v in step.grads.views])
        assert float((views - want).norm()) <= 1e-5 * float(want.norm()), mode
    assert step.replays == 5 and len(step._recorded) == 2     # nothing was recorded again


def test_resume_from_a_reference_style_checkpoint(tmp_path, fgcn_math):
    """Two steps, a checkpoint in the layout of the reference's CheckpointManager.save_checkpoint (progress.py:209-226: one
    state_dict per object of ``state_dict_objects`` + "epoch", torch.save), fresh objects, load, two more steps == four steps."""
    from fusion_gcn_amd.optim import create_optimizer
    from fusion_gcn_amd.session.procedures import DefaultBatchProcessor, GraphStep
    from fusion_gcn_amd.session.session import Session
    shape, classes = (1, 24, 20, 3), 27
    data = batches([4, 4, 4, 4], shape, classes)
    base = agcn(shape, classes)

    def objects(model):
        opt = create_optimizer("ADAM", model, 1e-3, weight_decay=0.01)          # config/utd-mhad/skeleton/agcn.yaml:15-22
        sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=3)
        proc = DefaultBatchProcessor(GraphStep())
        held = {"model": model, "optimizer": opt, "lr_scheduler": sched}
        proc.get_state_dict_objects(held)
        return opt, sched, proc, held

    def run(model, opt, sched, proc, part):
        for b in part:                                                         # one "epoch" per batch: the scheduler steps per epoch
            Session.train_epoch(proc, model, F.cross_entropy, [b], opt)
            sched.step()

    straight = copy.deepcopy(base).to(DEV)
    run(straight, *objects(straight)[:3], data)

    first = copy.deepcopy(base).to(DEV)
    opt, sched, proc, held = objects(first)
    run(first, opt, sched, proc, data[:2])
    torch.save({**{k: v.state_dict() for k, v in held.items()}, "epoch": 1}, tmp_path / "checkpoint_1_0.5.pt")

    resumed = copy.deepcopy(base).to(DEV)
    opt, sched, proc, held = objects(resumed)
    cp = torch.load(tmp_path / "checkpoint_1_0.5.pt", weights_only=False)
    for name, obj in held.items():
        obj.load_state_dict(cp[name])
    assert cp["epoch"] == 1 and opt.steps == 2
    run(resumed, opt, sched, proc, data[2:])
    torch.cuda.synchronize()
    sa, sb = resumed.state_dict(), straight.state_dict()
    for k in sa:
        if sa[k].is_floating_point():
            assert float((sa[k] - sb[k]).abs().max()) <= 1e-6 * max(1e-3, float(sb[k].abs().max())), k
        else:
            assert torch.equal(sa[k], sb[k]), k
    assert opt.param_groups[0]["lr"] == pytest.approx(sched.get_last_lr()[0])


def test_graph_step_in_the_bf16_mode_is_the_mixed_precision_step():
    """``GraphStep(math="bf16")`` (what ``get_batch_processor_from_config`` builds for --mixed_precision + hip_graph) trains like the eager
    ``MixedPrecisionStep``: both passes of every step in the bf16 math mode, whatever mode is current around them."""
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.session.procedures import DefaultBatchProcessor, GraphStep, MixedPrecisionStep
    shape, classes = (1, 24, 20, 3), 27
    data = batches([4, 4, 4], shape, classes)
    base = agcn(shape, classes)
    before = ops.get_math_mode()
    eager, keep_e = train(copy.deepcopy(base), DefaultBatchProcessor(MixedPrecisionStep()), data)
    graph, keep_g = train(copy.deepcopy(base), DefaultBatchProcessor(GraphStep(math="bf16")), data)
    assert ops.get_math_mode() == before
    assert_same_training(graph, eager, keep_g, keep_e)
    # and it is not the f32 training (the bf16 operands show in the loss from the first step on)
    f32, keep_f = train(copy.deepcopy(base), DefaultBatchProcessor(GraphStep(math="f32")), data)
    assert abs(float(keep_f.losses[0]) - float(keep_g.losses[0])) > 1e-6


def test_graph_step_with_dropout_draws_new_masks_per_replay():
    """nn.Dropout between the blocks (reference agcn.py:166-172): the recorded step is not compared with an eager one (other masks);
    every replay must draw its own mask (torch advances the generator offset of a recorded graph per replay)."""
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.session.procedures import GraphStep
    from fusion_gcn_amd.util import Graph
    shape, classes = (1, 24, 20, 3), 27
    model = Model(shape, classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint), dropout=0.3)
    filler.fill_state_dict(model.state_dict())
    model = model.to(DEV).train()
    (x, y, _), = batches([4], shape, classes)
    step = GraphStep()
    losses = []
    for _ in range(4):
        for p in model.parameters():
            p.grad = None
        _, loss = step.forward(model, F.cross_entropy, x, y)
        step.backward(loss)
        losses.append(float(loss))
    assert step.replays == 4 and len(step._recorded) == 1
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    assert len(set(losses)) > 1, losses               # different masks -> different losses


# ---- two data-parallel ranks through the harness (one device shared, gloo: the control flow of the N > 1 path, not a measurement) ----
def _dp_data(shape, classes):
    return batches([4, 4, 4], shape, classes, seed=11)


def _dp_worker(rank, world, port, out_path, mode):
    import os

    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fusion_gcn_amd import ops
        from fusion_gcn_amd.dp import FlatGradients, broadcast_parameters, shard_batch
        from fusion_gcn_amd.optim import FlatOptimizer
        from fusion_gcn_amd.session.procedures import DefaultBatchProcessor, GraphStep
        from fusion_gcn_amd.session.session import Session
        shape, classes = (1, 24, 20, 3), 27
        with ops.math_mode(mode):
            model = agcn(shape, classes).to(DEV).train()
            if rank:                                   # replicas start different on purpose: the broadcast must make them equal
                with torch.no_grad():
                    for p in model.parameters():
                        p.add_(0.01)
            broadcast_parameters(model, src=0)
            grads = FlatGradients(model.parameters())
            opt = FlatOptimizer(model.parameters(), "SGD", 0.01, momentum=0.9, grads=grads)
            step = GraphStep(grads=grads)
            mine = []
            for x, y, idx in _dp_data(shape, classes):
                sl = shard_batch(x.shape[0], rank, world)
                mine.append((x[sl].contiguous(), y[sl].contiguous(), idx[sl]))
            Session.train_epoch(DefaultBatchProcessor(step), model, F.cross_entropy, mine, opt)
            torch.cuda.synchronize()
            assert step.replays == 3 and len(step._recorded) == 1
        torch.save({k: v.cpu() for k, v in model.state_dict().items()}, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_two_ranks_through_graph_step_match_two_sequential_replicas(tmp_path, fgcn_math):
    import socket

    import torch.multiprocessing as mp
    from fusion_gcn_amd.dp import FlatGradients, shard_batch
    from fusion_gcn_amd.optim import FlatOptimizer
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = str(tmp_path / "rank")
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, out, fgcn_math)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = [torch.load(f"{out}.{r}") for r in range(world)]
    for k in got[0]:                                   # parameters identical on both ranks; BatchNorm statistics are per replica
        if "running_" not in k and "num_batches" not in k:
            assert torch.equal(got[0][k], got[1][k]), k

    # the same two shards one after the other in this process: per-replica BatchNorm, averaged gradients, the same update twice
    shape, classes = (1, 24, 20, 3), 27
    reps = []
    for _ in range(world):
        m = agcn(shape, classes).to(DEV).train()
        g = FlatGradients(m.parameters())
        reps.append((m, g, FlatOptimizer(m.parameters(), "SGD", 0.01, momentum=0.9, grads=g)))
    for x, y, _ in _dp_data(shape, classes):
        for r, (m, g, opt) in enumerate(reps):
            sl = shard_batch(x.shape[0], r, world)
            opt.zero_grad()
            F.cross_entropy(m(x[sl].contiguous()), y[sl]).backward()
            g.gather()
        mean = sum(g.flat for _, g, _ in reps) / world
        for m, g, opt in reps:
            g.flat.copy_(mean)
            opt.step()
    torch.cuda.synchronize()
    for r, (m, _, _) in enumerate(reps):
        want = m.state_dict()
        for k, v in got[r].items():
            w = want[k].cpu()
            if v.is_floating_point():
                assert float((v - w).abs().max()) <= 2e-6 * max(1e-3, float(w.abs().max())), (r, k)
            else:
                assert torch.equal(v, w), (r, k)
