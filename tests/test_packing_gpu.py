"""One-launch weight re-layout (fgcn_pack_run / fusion_gcn_amd.packing): every packed form of every block variant, in every math
mode, must equal the form built the long way -- torch cat / permute / pad of the reference-layout parameters
(torch_src/models/mmargcn/agcn.py:41-42,71-73,77 Conv2d weights) followed by the single-form layout kernels (ops.pack_k4 is a
pure torch re-layout, ops.pack_split3 the bit-exact three-way bf16 split) -- bit for bit, and a refresh after a parameter update
must equal a rebuild."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

VARIANTS = [("first", 3, 64, 1, "none", True), ("identity64", 64, 64, 1, "identity", False), ("down_s2", 64, 128, 2, "conv", True),
            ("identity256", 256, 256, 1, "identity", False), ("down_s2_256", 128, 256, 2, "conv", True)]


def _params(cfg, seed):
    from fusion_gcn_amd.block import param_names
    g = torch.Generator().manual_seed(seed)
    P = {}
    for n in param_names(cfg):
        if n == "gcn1.adj_b":
            shape = (3, 25, 25)
        elif n.endswith("bn.weight") or n.endswith("bn.bias") or n.endswith("down.1.weight") or n.endswith("down.1.bias"):
            shape = (cfg.cout,)
        elif ".conv_a." in n or ".conv_b." in n:
            shape = (cfg.ic, cfg.cin, 1, 1) if n.endswith("weight") else (cfg.ic,)
        elif n.startswith("tcn1.conv"):
            shape = (cfg.cout, cfg.cout, 9, 1) if n.endswith("weight") else (cfg.cout,)
        else:                                      # conv_d, down.0, residual.conv
            shape = (cfg.cout, cfg.cin, 1, 1) if n.endswith("weight") else (cfg.cout,)
        P[n] = torch.randn(shape, generator=g).cuda()
    return P


def _pad_last(t, n):
    return t if t.shape[-1] == n else F.pad(t, (0, n - t.shape[-1]))


def _long_way(P, cfg, mode):
    """the forms as torch ops + single-form layout kernels (the pre-plan construction)"""
    from fusion_gcn_amd import ops
    cin, cout, ic, cx = cfg.cin, cfg.cout, cfg.ic, cfg.cx
    x3 = mode in ops.SPLIT_MODES
    R = {}
    emb_t = _pad_last(torch.cat([P[f"gcn1.conv_{g}.{k}.weight"].view(ic, cin) for k in range(3) for g in "ab"], 0), cx)
    R["emb"] = emb_t.t().contiguous().unsqueeze(0)
    R["emb_t"] = emb_t.contiguous().unsqueeze(0)
    R["emb_b"] = torch.cat([P[f"gcn1.conv_{g}.{k}.bias"] for k in range(3) for g in "ab"])
    d_list = [_pad_last(P[f"gcn1.conv_d.{k}.weight"].view(cout, cin), cx) for k in range(3)]
    R["d"] = torch.cat([w.t() for w in d_list], 0).contiguous()
    R["d4"] = ops.pack_spatial(R["d"], cx)
    if mode in ("bf16x3", "bf16") and cx % 64 == 0:   # the tile form of the fused spatial kernel takes the plain split form (bf16: its part 0)
        R["d_s3"] = ops.pack_split3(R["d"].unsqueeze(0))
    R["d_t"] = torch.cat(d_list, 1).contiguous().unsqueeze(0)
    R["d_b"] = P["gcn1.conv_d.0.bias"] + P["gcn1.conv_d.1.bias"] + P["gcn1.conv_d.2.bias"]
    if cfg.has_down:
        down = _pad_last(P["gcn1.down.0.weight"].view(cout, cin), cx)
        R["down"], R["down_t"] = down.t().contiguous().unsqueeze(0), down.contiguous().unsqueeze(0)
    wt = P["tcn1.conv.weight"].view(cout, cout, -1)
    R["t"], R["t_t"] = wt.permute(2, 1, 0).contiguous(), wt.permute(2, 0, 1).contiguous()
    if cfg.stride == 1:
        R["t4"], R["t_t4"] = ops.pack_conv(R["t"]), ops.pack_conv(R["t_t"])
    else:
        for par, tag in ((0, "e"), (1, "o")):
            R[f"t_t4_{tag}"] = ops.pack_conv(R["t_t"][par::2].contiguous())
            if x3:
                R[f"t4_{tag}"] = ops.pack_conv(R["t"][par::2].contiguous())
    if cfg.residual == "conv":
        res = _pad_last(P["residual.conv.weight"].view(cout, cin), cx)
        R["res"], R["res_t"] = res.t().contiguous().unsqueeze(0), res.contiguous().unsqueeze(0)
    if x3:
        for key in ("emb", "emb_t", "d_t", "down", "down_t"):
            if key in R and R[key].shape[1] % 32 == 0:
                R[key + "_s3"] = ops.pack_split3(R[key])
    if mode in ("bf16x3", "f16x2", "bf16") and cx % 64 == 0 and cout % 64 == 0:   # the fused spatial backward's weights: three-way bf16 splits in every split mode
        R["d_t_b3"] = ops.pack_split3(R["d_t"])
    if mode in ops.SPLIT_MODES and cx % 64 == 0:   # the embedding backward in tile form (fgcn_emb_dx_tile)
        R["emb_t_b3"] = ops.pack_split3(R["emb_t"])
    if mode in ops.SPLIT_MODES and cx % 32 == 0:   # the embedding forward with the gram on chip (fgcn_emb_fwd_tile)
        R["emb_b3"] = ops.pack_split3(R["emb"])
    return R


def _same(a, b):
    if a.dtype == torch.bfloat16:
        return a.shape == b.shape and torch.equal(a.view(torch.int16), b.view(torch.int16))
    return a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
@pytest.mark.parametrize("name,cin,cout,stride,residual,has_down", VARIANTS)
def test_every_form_equals_the_long_way(mode, name, cin, cout, stride, residual, has_down):
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.block import BlockConfig, pack_weights
    cfg = BlockConfig(cin=cin, cout=cout, stride=stride, residual=residual, has_down=has_down)
    P = _params(cfg, seed=hash(name) % 1000)
    with ops.math_mode(mode):
        W = pack_weights(P, cfg)
        want = _long_way(P, cfg, mode)
        assert set(W.specs) == set(want), (sorted(W.specs), sorted(want))
        for key, ref in want.items():
            assert key in W
            assert _same(W[key], ref), (mode, name, key, tuple(W[key].shape), tuple(ref.shape))
        # an optimizer step changes the parameters in place: ONE launch refreshes every live form
        ptrs = {k: W[k].data_ptr() for k in want}
        with torch.no_grad():
            for p in P.values():
                p.mul_(1.5).add_(0.25)
        W.refresh()
        want2 = _long_way(P, cfg, mode)
        for key, ref in want2.items():
            assert W[key].data_ptr() == ptrs[key]              # buffers keep their addresses (HIP-graph capture)
            assert _same(W[key], ref), (mode, name, key, "after refresh")


def test_model_refreshes_all_blocks_in_one_launch_and_tracks_parameter_versions():
    """Model.forward re-packs only when a parameter version changed, through ONE plan over all blocks; logits follow the
    parameters (a stale form would keep the old logits)."""
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.mmargcn.agcn import Model, SpatialTemporalConv
    from fusion_gcn_amd.util import Graph
    from oracle import filler
    dev = torch.device("cuda:0")
    shape = (2, 1, 16, 20, 3)
    model = Model(shape[1:], 27, Graph(utd.skeleton_edges, center_joint=utd.center_joint))
    filler.fill_state_dict(model.state_dict())
    model = model.to(dev).eval()
    x = torch.from_numpy(filler.skeleton_input("x.pack", shape)).float().to(dev)
    with torch.no_grad():
        a = model(x).clone()
        assert getattr(model, "_pack_plan", None) is None       # first forward: the blocks built their forms lazily
        b = model(x).clone()
        assert torch.equal(a, b) and getattr(model, "_pack_plan", None) is None      # nothing changed: nothing re-packed
        for p in model.parameters():
            if p.dim() == 4:
                p.mul_(1.01)                                    # in place: bumps the version counters
        c = model(x).clone()
        plan = model._pack_plan[1]
        blocks = [m for m in model.modules() if isinstance(m, SpatialTemporalConv)]
        assert len(plan.forms) == sum(len(blk._wcache[1].live) for blk in blocks) and plan.n_wg > 0
        assert not torch.equal(a, c)
        d = model(x).clone()
        assert torch.equal(c, d) and model._pack_plan[1] is plan
    # the same weights through freshly built forms give the same logits
    fresh = Model(shape[1:], 27, Graph(utd.skeleton_edges, center_joint=utd.center_joint)).to(dev).eval()
    fresh.load_state_dict(model.state_dict())
    with torch.no_grad():
        assert torch.equal(fresh(x), c)
