"""Model-level gradient parity with ReLU-decision accounting (north star: <= 1e-3 rel vs the CPU reference).

The flat gradient of the 10-block model against the float64 oracle (oracle/agcn_oracle.py, pinned to the reference's own
outputs by tests/test_oracle_golden.py), at BASELINE configs[0] (cfg1) and the config-2 fixture shape:

  (i)   ReLU decisions of the HIP forward vs the oracle's, per block (20 ReLU layers);
  (ii)  HIP backward gated on the ORACLE's decisions (written into the saved one-bit sign images): <= 1e-4 -- what is left
        is arithmetic only, so a real backward bug at model scale cannot hide behind "ReLU flips";
  (iii) the un-injected error is bounded by what the counted flips explain.

Reference: autograd over torch_src/models/mmargcn/agcn.py:183-200; floor of the reference against itself (fp32 vs fp64 on
these two cases): tests/golden/model.npz ``*.ref_f32_vs_f64_grad_rel`` = 8.5e-5 / 2.5e-4."""
import math

import numpy as np
import pytest
import torch

from oracle import filler
from oracle import relu_masks as RM

pytestmark = pytest.mark.gpu

INJECTED_TOL = 1e-4          # arithmetic-only gradient error of the whole model
FLIP_GAIN = 2.0              # un-injected error <= INJECTED_TOL + FLIP_GAIN * sqrt(flips / decisions per ReLU layer)
# measured on MI355X (gpurun_out r02): cfg1 4 flips of 5.6 M decisions -> 1.9e-3 as is, 9.6e-7 injected; the smoke case ONE flip
# of 2.25 M -> 3.5e-3 as is, 1.2e-6 injected: one flipped element of a 1e5-element layer is sqrt(1e-5) = 3e-3 of the gradient


def _case(tag, shape, classes, gname):
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from fusion_gcn_amd.datasets.utd_mhad import constants as utd
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.util import Graph
    c = {"utd": utd, "ntu": ntu}[gname]
    n, m, t, v, ch = shape
    model = Model((m, t, v, ch), classes, Graph(c.skeleton_edges, center_joint=c.center_joint))
    filler.fill_state_dict(model.state_dict())
    x = torch.from_numpy(filler.skeleton_input(f"x.{tag}", shape, empty_second_body=(m > 1)))
    labels = torch.from_numpy(filler.uniform(f"y.{tag}", (n,), 0, classes).astype(np.int64))
    return model, x, labels


def flip_bound(flips: int, decisions: int, layers: int = 20) -> float:
    """Error a correct float32 backward may show when ``flips`` of ``decisions`` ReLU decisions differ from the oracle's: each
    flipped element removes / adds one upstream gradient element of typical magnitude, so the relative error grows like
    sqrt(flip fraction of one layer) (the reference's own fp32 run against its fp64 run shows the
    same jumps, 12 flips at N = 16 -> 7.3e-4).  With no flip the bound is the arithmetic-only 1e-4."""
    return INJECTED_TOL + FLIP_GAIN * math.sqrt(flips / (decisions / layers))


@pytest.mark.parametrize("tag,shape,classes,gname", [("cfg1", (2, 1, 100, 20, 3), 27, "utd"),
                                                     ("cfg2_small", (2, 2, 32, 25, 3), 60, "ntu")])
def test_flat_gradient_with_relu_flip_accounting(golden, tag, shape, classes, gname, fgcn_math):
    dev = torch.device("cuda:0")
    model, x, labels = _case(tag, shape, classes, gname)
    sd = {k: (v.detach().double().clone() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    rep = RM.gradient_parity_report(model, x.float().to(dev), labels.to(dev), x.double(), labels, sd)
    ref_floor = float(golden("model.npz")[f"{tag}.ref_f32_vs_f64_grad_rel"])
    frac = rep["flips"] / rep["decisions"]
    print(f"[{tag} {fgcn_math}] logits {rep['logits_err']:.2e} loss {rep['loss_err']:.2e} | ReLU flips {rep['flips']} of "
          f"{rep['decisions']} ({frac:.2e}) per block (g, o): {rep['flips_per_block']} | flat-grad rel-L2: plain "
          f"{rep['err_plain']:.2e}, oracle decisions injected {rep['err_injected']:.2e} "
          f"(reference fp32-vs-fp64 on this case: {ref_floor:.2e})")
    assert rep["logits_err"] < 1e-5 and rep["loss_err"] < 1e-5
    # (i) flips are rare: only pre-activations within float32 rounding of zero can flip -- and no more frequent than in the
    # reference's OWN float32 run against its float64 run on this case (tests/golden/model.npz, written by oracle/gen_golden.py from
    # hooks on the imported reference: 1 / 2 flips).  Counts this small are Poisson noise, hence 3 x + 3
    ref_flips = int(golden("model.npz")[f"{tag}.ref_f32_vs_f64_relu_flips"].sum())
    assert int(golden("model.npz")[f"{tag}.relu_decisions"].sum()) == rep["decisions"]
    print(f"    ReLU flips vs fp64: HIP {rep['flips']}, the reference's own float32 run {ref_flips}")
    assert rep["flips"] <= 3 * ref_flips + 3, (rep["flips"], ref_flips, rep["flips_per_block"])
    assert frac < 2e-5, rep["flips_per_block"]
    # (ii) arithmetic-only error
    assert rep["err_injected"] < INJECTED_TOL, rep
    # (iii) the rest is explained by the counted flips
    assert rep["err_plain"] <= flip_bound(rep["flips"], rep["decisions"]), rep
    # the north star's 1e-3 is met with a factor 100 to spare once the decisions agree; without a flip it is met as is
    if rep["flips"] == 0:
        assert rep["err_plain"] < INJECTED_TOL


def test_headline_shape_against_the_oracle_at_full_size():
    """BASELINE configs[1] at its full (C,T,V,M) = (3,300,25,2) and 60 classes, 16 clips (what the CPU oracle affords: ~1 min of
    float64 work on the box's cores): HIP logits / loss against the float64 oracle, the flat gradient of all 3.47 M parameters with
    ReLU-flip accounting -- the same three-part statement as above, at the size the benchmark runs (bench.py reports the same
    comparison against its float32 oracle run as ``parity_at_full_shape``)."""
    import os
    from fusion_gcn_amd import ops
    n = int(os.environ.get("FGCN_FULL_SHAPE_CLIPS", "16"))
    dev = torch.device("cuda:0")
    model, x, labels = _case("full", (n, 2, 300, 25, 3), 60, "ntu")
    with torch.no_grad():                       # O(1) BatchNorm scale in the graph convolutions, as the benchmark sets it
        for k, p in model.named_parameters():
            if k.endswith("gcn1.bn.weight"):
                p.fill_(1.0)
    sd = {k: (v.detach().double().clone() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    oracle = RM.oracle_side(x.double(), labels, sd, [k for k, _ in model.named_parameters()])
    del sd
    model = model.to(dev).train()
    for mode in ("bf16x3", "f16x2", "f32"):
        with ops.math_mode(mode):
            rep = RM.gradient_parity_report(model, x.float().to(dev), labels.to(dev), oracle=oracle)
        frac = rep["flips"] / rep["decisions"]
        print(f"[full shape, {n} clips, {mode}] logits {rep['logits_err']:.2e} loss {rep['loss_err']:.2e} | ReLU flips {rep['flips']} of "
              f"{rep['decisions']} ({frac:.2e}) | flat-grad rel-L2: plain {rep['err_plain']:.2e}, oracle decisions injected "
              f"{rep['err_injected']:.2e}")
        assert rep["logits_err"] < 1e-5 and rep["loss_err"] < 1e-5, rep
        assert frac < 2e-6, rep["flips_per_block"]          # ~3e-7 of the ReLU inputs in the reference's own runs
        assert rep["err_injected"] < INJECTED_TOL, rep
        assert rep["err_plain"] <= flip_bound(rep["flips"], rep["decisions"]), rep


@pytest.mark.parametrize("tag,dataset,joints,classes,n_imu", [("ntu27_T300", "ntu", 27, 60, 2), ("mmact22_T300", "mmact", 22, 35, 4)])
def test_other_baseline_shapes_against_the_oracle_at_full_length(tag, dataset, joints, classes, n_imu):
    """BASELINE configs 3 and 4 at their full T = 300 (NTU graph + 2 IMU joints, V = 27; MMAct COCO-18 graph + 4 IMU joints, V = 22;
    M = 2), 4 clips -- the fixture-size comparison of tests/test_block_model_gpu.py::test_other_baseline_shapes_vs_reference runs T = 16:
    logits / loss against the float64 oracle, the flat gradient with ReLU-flip accounting, in the three float32-class math modes.  The
    network is built on the fused skeleton + IMU graph the mmargcn mode builds (models/mmargcn/fusion.py:52-65; reference
    fusion.py:65-89); the IMU joints carry uniform [0, 1) signals like the reference's min-max normalised ones."""
    import os
    from fusion_gcn_amd import ops
    from fusion_gcn_amd.datasets.mmact import constants as mmact
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.models.mmargcn.fusion import get_skeleton_imu_fusion_graph
    from fusion_gcn_amd.util import Graph
    c = {"ntu": ntu, "mmact": mmact}[dataset]
    n = int(os.environ.get("FGCN_FULL_LENGTH_CLIPS", "4"))
    dev = torch.device("cuda:0")
    graph = get_skeleton_imu_fusion_graph(Graph(c.skeleton_edges, center_joint=c.center_joint), "append_center", n_imu)
    shape = (n, 2, 300, joints, 3)
    model = Model(shape[1:], classes, graph)
    filler.fill_state_dict(model.state_dict())
    with torch.no_grad():                       # O(1) BatchNorm scale in the graph convolutions, as the benchmark sets it
        for k, p in model.named_parameters():
            if k.endswith("gcn1.bn.weight"):
                p.fill_(1.0)
    x = torch.from_numpy(filler.skeleton_input(f"x.{tag}", shape, empty_second_body=True))
    x[..., joints - n_imu:, :] = torch.from_numpy(filler.uniform(f"imu.{tag}", (n, 2, 300, n_imu, 3), 0, 1)).to(x.dtype)
    labels = torch.from_numpy(filler.uniform(f"y.{tag}", (n,), 0, classes).astype(np.int64))
    sd = {k: (v.detach().double().clone() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    oracle = RM.oracle_side(x.double(), labels, sd, [k for k, _ in model.named_parameters()])
    del sd
    model = model.to(dev).train()
    for mode in ("bf16x3", "f16x2", "f32"):
        with ops.math_mode(mode):
            rep = RM.gradient_parity_report(model, x.float().to(dev), labels.to(dev), oracle=oracle)
        frac = rep["flips"] / rep["decisions"]
        print(f"[{tag}, {n} clips, {mode}] logits {rep['logits_err']:.2e} loss {rep['loss_err']:.2e} | ReLU flips {rep['flips']} of "
              f"{rep['decisions']} ({frac:.2e}) | flat-grad rel-L2: plain {rep['err_plain']:.2e}, oracle decisions injected "
              f"{rep['err_injected']:.2e}")
        assert rep["logits_err"] < 1e-5 and rep["loss_err"] < 1e-5, rep
        assert frac < 2e-6, rep["flips_per_block"]
        assert rep["err_injected"] < INJECTED_TOL, rep
        assert rep["err_plain"] <= flip_bound(rep["flips"], rep["decisions"]), rep
