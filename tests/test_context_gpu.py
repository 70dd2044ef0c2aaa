"""Library contexts (include/fgcn.h: fgcn_ctx_*; fusion_gcn_amd.ops.Context): the settings every launcher reads -- math mode, product
form, kernel-variant table -- are current PER THREAD, so two Python threads running two models in two math modes on two streams
produce exactly what each mode produces alone ("stateless and re-entrant")."""
import threading

import numpy as np
import pytest
import torch


def test_context_api_without_a_gpu():
    """Create / make current / read back / destroy; a setter changes the calling thread's current context only; another thread that
    has no context of its own keeps reading the process-wide defaults."""
    import ctypes

    from fusion_gcn_amd import _lib, ops
    lib = _lib.load()
    assert ops.current_context() is not None and lib.fgcn_ctx_get_current() is None
    base_mode = ops.get_math_mode()
    with ops.context("bf16") as ctx:
        assert lib.fgcn_ctx_get_current() == ctx.handle and ops.get_math_mode() == "bf16"
        assert lib.fgcn_set_tuning(7, 2) == 0 and lib.fgcn_get_tuning(7) == 2
        seen = {}

        def other():
            seen["mode"], seen["key7"], seen["cur"] = ops.get_math_mode(), lib.fgcn_get_tuning(7), lib.fgcn_ctx_get_current()
            with ops.context("f16x2"):
                seen["inner"] = ops.get_math_mode()
            seen["after"] = ops.get_math_mode()
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen == {"mode": base_mode, "key7": 0, "cur": None, "inner": "f16x2", "after": base_mode}
        assert ops.get_math_mode() == "bf16"                           # the other thread's context did not leak into this one
        # a context that is current on this thread cannot be destroyed
        assert lib.fgcn_ctx_destroy(ctx.handle) != 0
        with ops.context() as nested:                                   # a copy of the present settings
            assert ops.get_math_mode() == "bf16" and lib.fgcn_get_tuning(7) == 2 and nested.handle != ctx.handle
            ops.set_math_mode("f32")
        assert ops.get_math_mode() == "bf16"
    assert lib.fgcn_ctx_get_current() is None and ops.get_math_mode() == base_mode and lib.fgcn_get_tuning(7) == 0
    bad = ctypes.c_int(lib.fgcn_set_tuning(32, 1)).value
    assert bad != 0                                                     # misuse is an error code, never a silent no-op
    with pytest.raises(_lib.FgcnError):
        _lib.check(bad, "fgcn_set_tuning")


def _model_and_batch():
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.util import Graph
    from oracle import filler
    shape, classes = (2, 1, 24, 20, 3), 27
    model = Model(shape[1:], classes, Graph(utd.skeleton_edges, center_joint=utd.center_joint), num_layers=4)
    filler.fill_state_dict(model.state_dict())
    x = torch.from_numpy(filler.skeleton_input("x.ctx", shape)).float()
    y = torch.from_numpy(filler.uniform("y.ctx", (shape[0],), 0, classes).astype(np.int64))
    return model, x, y


def test_path_options_are_per_context_without_a_gpu():
    """The kernel-form options (fusion_gcn_amd/paths.py) travel with the context: a fresh context starts from a COPY of the thread's
    present ones, changes stay inside it, another thread keeps the process defaults; FGCN_PATHS-style strings parse strictly."""
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.paths import PathOptions
    base = ops.paths()
    assert base is ops.current_context().paths and base.emb_tile and base.emb_tile_max_cin["bf16x3"] == 128 and not base.fuse_g
    with ops.context() as ctx:
        assert ctx.paths is not base and ctx.paths == base              # a copy, equal in value
        ctx.paths.emb_tile = False
        ctx.paths.fused_agg_wgrad_max_cout["bf16x3"] = 64
        assert ops.paths() is ctx.paths and not ops.paths().emb_tile
        seen = {}
        t = threading.Thread(target=lambda: seen.update(emb=ops.paths().emb_tile, same=ops.paths() is base))
        t.start()
        t.join()
        assert seen == {"emb": True, "same": True}                      # the other thread reads the process defaults
        with ops.context() as inner:                                    # nested: a copy of the CURRENT (changed) options
            assert not inner.paths.emb_tile and inner.paths.fused_agg_wgrad_max_cout["bf16x3"] == 64
            inner.paths.emb_tile = True
        assert not ops.paths().emb_tile
    assert ops.paths() is base and base.emb_tile and base.fused_agg_wgrad_max_cout["bf16x3"] == 128
    o = PathOptions().update_from("emb_tile=0, spatial_tile_min_cout=64,fused_agg_wgrad_max_cout=256,mix_vw_order=1:2,fuse_g=1,"
                                  "bn_sums_in_dgrad.bf16x3=0, emb_tile_max_cin.bf16=64")
    assert (not o.emb_tile and set(o.spatial_tile_min_cout.values()) == {64} and o.fuse_g and o.mix_vw_order == (1, 2)
            and set(o.fused_agg_wgrad_max_cout.values()) == {256} and o.get("bn_sums_in_dgrad", "bf16x3") is False
            and o.get("bn_sums_in_dgrad", "f32") is True and o.emb_tile_max_cin == {"f32": 128, "bf16": 64, "bf16x3": 128, "f16x2": 128})
    assert PathOptions().get("bn_sums_in_dgrad", "bf16") is False and PathOptions().get("spatial_tile_min_cout", "bf16") == 64
    with pytest.raises(ValueError, match="no entry for math mode"):
        PathOptions().update_from("emb_tile.bf16=0")
    with pytest.raises(ValueError, match="unknown path option"):
        PathOptions().update_from("emb_tiles=0")


def _run(mode, steps, stream, out, barrier=None, paths=None, key=None):
    """`steps` forward + backward passes of a fresh copy of the model in `mode`, inside a context of its own, on `stream`; ``paths``:
    kernel-form options set on that context (a dict of PathOptions fields)."""
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.loss import cross_entropy
    dev = torch.device("cuda:0")
    model, x, y = _model_and_batch()
    model = model.to(dev).train()
    x, y = x.to(dev), y.to(dev)
    res = []
    try:
        with ops.context(mode) as c, torch.cuda.stream(stream):
            for name, value in (paths or {}).items():
                assert hasattr(c.paths, name), name
                setattr(c.paths, name, value)
            for _ in range(steps):
                if barrier is not None:
                    barrier.wait(timeout=120)                           # both threads enter every step together
                model.zero_grad(set_to_none=True)
                logits = model(x)
                loss = cross_entropy(logits, y)
                loss.backward()
                stream.synchronize()
                res.append((logits.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters()]).clone()))
        out[key or mode] = res
    except BaseException as e:      # noqa: BLE001 - the other thread must not wait for a partner that is gone
        out[key or mode] = e
        if barrier is not None:
            barrier.abort()


@pytest.mark.gpu
def test_two_threads_in_two_math_modes_match_the_single_mode_runs():
    from fusion_gcn_amd import ops
    steps = 3
    alone = {}
    for mode in ("f32", "bf16x3"):
        _run(mode, steps, torch.cuda.Stream(), alone)
    assert not torch.equal(alone["f32"][0][1], alone["bf16x3"][0][1])      # the two modes really differ in the last bits
    before = ops.get_math_mode()
    both, barrier = {}, threading.Barrier(2)
    threads = [threading.Thread(target=_run, args=(mode, steps, torch.cuda.Stream(), both, barrier)) for mode in ("f32", "bf16x3")]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    for mode in ("f32", "bf16x3"):
        if isinstance(both.get(mode), BaseException):
            raise both[mode]
    assert ops.get_math_mode() == before
    for mode in ("f32", "bf16x3"):
        for (la, ga), (lb, gb) in zip(alone[mode], both[mode]):
            assert torch.equal(la, lb) and torch.equal(ga, gb), mode       # bit for bit what the mode produces alone


@pytest.mark.gpu
def test_two_threads_with_different_path_options_match_their_single_runs():
    """Two models in ONE math mode but with different kernel-form options -- one on the fused tile kernels (the defaults), one on the
    unfused chains of the embedding and spatial backward -- on two threads and two streams: each reproduces, bit for bit, what it
    produces alone, and the two really took different kernels (their gradients differ in the last bits).  Before round 6 these
    switches were module globals of block.py and the two models could not differ."""
    from fusion_gcn_amd import ops
    steps = 2
    unfused = {"emb_tile": False, "emb_fwd_tile": False, "spatial_bwd_tile": False, "spatial_wgrad_tile": False, "bn_sums_in_dgrad": False}
    alone = {}
    _run("bf16x3", steps, torch.cuda.Stream(), alone, key="tile")
    _run("bf16x3", steps, torch.cuda.Stream(), alone, paths=unfused, key="unfused")
    for k in ("tile", "unfused"):
        if isinstance(alone[k], BaseException):
            raise alone[k]
    assert not torch.equal(alone["tile"][0][1], alone["unfused"][0][1])          # other kernels: other rounding
    err = float((alone["tile"][0][1] - alone["unfused"][0][1]).norm() / alone["tile"][0][1].norm())
    assert err < 1e-4, err                                                     # ... of the same gradient
    both, barrier = {}, threading.Barrier(2)
    threads = [threading.Thread(target=_run, args=("bf16x3", steps, torch.cuda.Stream(), both, barrier, None, "tile")),
               threading.Thread(target=_run, args=("bf16x3", steps, torch.cuda.Stream(), both, barrier, unfused, "unfused"))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    for k in ("tile", "unfused"):
        if isinstance(both.get(k), BaseException):
            raise both[k]
        for (la, ga), (lb, gb) in zip(alone[k], both[k]):
            assert torch.equal(la, lb) and torch.equal(ga, gb), k
    assert ops.paths().emb_tile and ops.paths().spatial_bwd_tile              # the process defaults were never touched


@pytest.mark.gpu
def test_backward_runs_in_the_forwards_context():
    """The forward's context is what the backward (on the autograd thread) computes in: leaving the context between the two changes
    nothing, and the thread's own mode stays what it was."""
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.loss import cross_entropy
    dev = torch.device("cuda:0")
    model, x, y = _model_and_batch()
    model = model.to(dev).train()
    x, y = x.to(dev), y.to(dev)
    grads = {}
    for leave in (False, True):
        model.zero_grad(set_to_none=True)
        with ops.math_mode("f32"):
            with ops.context("bf16x3"):
                loss = cross_entropy(model(x), y)
                if not leave:
                    loss.backward()
            if leave:
                loss.backward()                                        # the thread is back in f32; the Functions remember bf16x3
            assert ops.get_math_mode() == "f32"
        grads[leave] = torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
    assert torch.equal(grads[False], grads[True])
