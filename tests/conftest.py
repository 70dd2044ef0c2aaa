import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "math_modes(*modes): the float32-class math modes this test's kernel exists in (a subset of f32 / bf16x3 / "
                                       "f16x2): the module-wide parametrisation over the three modes is cut down to them, so a kernel that has no "
                                       "form in a mode is not collected there instead of showing up as a skip")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible, so a plain `pytest tests` works anywhere."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Every skipped test with its reason, by test id, as a file: the driver's `-q` tail shows a count only.  Written to
    $FGCN_SKIP_REPORT, or to gpurun_out/gpu_skips.txt when the run is on a GPU box (copied to profiles/rNN_gpu_skips.txt per round)."""
    path = os.environ.get("FGCN_SKIP_REPORT")
    if not path:
        try:
            import torch
            if not torch.cuda.is_available():
                return
        except Exception:  # pragma: no cover
            return
        path = os.path.join(ROOT, "gpurun_out", "gpu_skips.txt")
    skipped = terminalreporter.stats.get("skipped", [])
    by_reason = {}
    for rep in skipped:
        reason = rep.longrepr[2] if isinstance(rep.longrepr, tuple) else str(rep.longrepr)
        by_reason.setdefault(reason.replace("Skipped: ", ""), []).append(rep.nodeid)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            counts = {k: len(terminalreporter.stats.get(k, [])) for k in ("passed", "failed", "skipped", "error")}
            f.write(f"# pytest {' '.join(config.invocation_params.args)}: {counts}\n")
            for reason in sorted(by_reason, key=lambda r: -len(by_reason[r])):
                f.write(f"\n## {len(by_reason[reason])} x {reason}\n")
                for nid in by_reason[reason]:
                    f.write(f"{nid}\n")
    except OSError:  # pragma: no cover  (a read-only tree must not fail the run)
        pass


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))


# The kernel / block / model parity modules run three times on the GPU: in "f32" (exact f32 MFMA), in "bf16x3"
# (f32-accurate split-bf16 products, include/fgcn.h) and in "f16x2" (bf16x3 whose temporal / 1x1 convolutions form their products
# from block-scaled two-way f16 splits, FGCN_PRODUCTS_F16X2) -- the same oracle, the same tolerances.
BOTH_MATH_MODES = {"test_kernels_gpu", "test_block_model_gpu", "test_train_e2e_gpu", "test_imu_gcn", "test_grad_parity_gpu", "test_msg3d", "test_session_gpu"}


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in BOTH_MATH_MODES and "fgcn_math" in metafunc.fixturenames:
        only = metafunc.definition.get_closest_marker("math_modes")
        modes = [m for m in ("f32", "bf16x3", "f16x2") if only is None or m in only.args]
        metafunc.parametrize("fgcn_math", modes, indirect=True)


@pytest.fixture(autouse=True)
def fgcn_math(request):
    """Every test runs in an explicitly selected math mode: "f32" (exact-f32 MFMAs, the kernel tests' tight tolerances) unless the
    module is parametrised over the three float32-class modes.  (The LIBRARY's own default is bf16x3 -- tests/test_abi.py checks it.)"""
    mode = getattr(request, "param", "f32")
    from fusion_gcn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):           # no library built: only tests that never touch it can pass anyway
        yield mode
        return
    from fusion_gcn_amd import ops
    with ops.math_mode(mode):
        yield mode
