"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol include/fgcn.h
declares, and rejects malformed calls on the host before any launch (no GPU needed: validation precedes HIP)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from fusion_gcn_amd import _lib, build


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fgcn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fgcn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libfgcn.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signature table and header disagree"


def test_version_and_error_text(lib):
    assert lib.fgcn_version() >= 100
    rc = lib.fgcn_rows_gemm(None, None, None, None, None, 1, 1, 1, 1, 1, 4, 4, 4, _lib.TMap(1, 1, 0, 0, 1), 0, None)
    assert rc == -1 and b"null pointer" in lib.fgcn_last_error()
    with pytest.raises(_lib.FgcnError, match="null pointer"):
        _lib.check(rc, "fgcn_rows_gemm")


def test_host_side_validation(lib):
    buf = (C.c_float * 64)()
    p = C.addressof(buf)
    p16 = (p + 15) // 16 * 16
    ident = _lib.TMap(1, 1, 0, 0, 1)
    # N not a multiple of 4
    assert lib.fgcn_rows_gemm(p16, p16, p16, None, None, 1, 1, 1, 1, 4, 3, 4, 4, ident, 0, None) == -2
    # row stride smaller than the channel window
    assert lib.fgcn_rows_gemm(p16, p16, p16, None, None, 1, 1, 1, 1, 8, 4, 4, 4, ident, 0, None) == -1
    # misaligned base pointer
    assert lib.fgcn_rows_gemm(p16 + 4, p16, p16, None, None, 1, 1, 1, 1, 4, 4, 4, 4, ident, 0, None) == -2
    # bad temporal map
    assert lib.fgcn_rows_gemm(p16, p16, p16, None, None, 1, 1, 1, 1, 4, 4, 4, 4, _lib.TMap(0, 1, 0, 0, 1), 0, None) == -1
    # more joints than the kernels' 32-wide joint tile
    item = (_lib.MixItem * 1)()
    assert lib.fgcn_joint_mix(p16, p16, p16, 1, 1, 33, 4, 4, 4, 4, 1, 0, item, 1, 0, None) == -1
    assert b"bad B/T/V" in lib.fgcn_last_error()
    # spatial kernel: more than 256 channels
    assert lib.fgcn_spatial_fwd(p16, p16, p16, None, p16, None, 1, 1, 25, 512, 64, 512, 64, 3, 1, None) == -1
    assert lib.fgcn_spatial_fwd(p16, p16, p16, None, p16, None, 1, 1, 25, 3, 64, 4, 64, 3, 1, None) == -2   # Cin % 4
    # bn reduce with a wrong tile count
    assert lib.fgcn_bn_act_bwd_reduce(p16, p16, None, p16, p16, None, None, p16, 7, 1000, 64, 0, 1, None) == -1   # needs 16 tiles
    assert lib.fgcn_elem_tiles(1000) == 16 and lib.fgcn_elem_tiles(10 ** 7) == 1024 and lib.fgcn_rows_gemm_tiles(129) == 2
    assert lib.fgcn_spatial_tiles(128, 300) == 128 * 10
    # typed entry points (half-precision activation storage): a mask the kernel is not built for, and bfloat16 tensors outside math mode bf16
    assert lib.fgcn_bn_act_t(p16, p16, None, None, p16, None, 16, 8, 0, 1, 8, None) == -1                  # bit 3 does not exist
    assert b"half_mask" in lib.fgcn_last_error()
    assert lib.fgcn_tconv_halo_t(p16, p16, p16, None, None, 1, 4, 25, 32, 32, 32, 32, 4, 1, 0, 4, 4, 1, 0, 9, 1, -4, 2, None) == -1   # out without in
    assert lib.fgcn_spatial_bwd_tile_t(p16, p16, p16, p16, p16, p16, 1, 4, 25, 64, 64, 64, 64, 64, 1, 0, None, 0, None, None, None, 5, None) == -1
    assert lib.fgcn_emb_dx_tile_t(p16, p16, p16, p16, p16, 1, 4, 25, 16, 64, 96, 64, 1, 0, None, 2, None) == -1
    mode = lib.fgcn_get_math_mode()
    lib.fgcn_set_math_mode(2)          # bf16x3: bfloat16 tensors are refused before anything is launched
    try:
        assert lib.fgcn_bn_act_t(p16, p16, None, None, p16, None, 16, 8, 0, 1, 1, None) == -1
        assert b"math mode bf16" in lib.fgcn_last_error()
        assert lib.fgcn_bn_act_bwd_apply_t(p16, 0, None, p16, p16, p16, None, None, p16, p16, None, 16, 8, 0, 1, 1, 0, 3, None) == -1
    finally:
        lib.fgcn_set_math_mode(mode)


def test_tile_kernel_geometry_and_tuning_keys(lib):
    """Host-side geometry of the two round-4 tile kernels (no launch): one workgroup per CU at the headline shapes, fewer when the
    batch is small, zero slabs / segments for sizes the kernels do not take; fgcn_set_tuning takes keys 0..31 and says so otherwise
    (keys 16..18 had been rejected unnoticed)."""
    B, V = 128, 25
    for T, cin, cout, combos in ((300, 64, 64, 1), (300, 64, 128, 1), (150, 128, 128, 1), (150, 128, 256, 2), (75, 256, 256, 4)):
        slabs = lib.fgcn_spatial_wgrad_tile_slabs(B, T, V, cin, cout)
        assert 0 < slabs * combos <= 256 and slabs * combos > 192, (T, cin, cout, slabs)   # one round of (almost) 256 workgroups
        segs = lib.fgcn_spatial_bwd_tile_segments(B, T, V)
        assert 0 < B * segs <= 256 and B * segs >= 128, (T, segs)
    assert lib.fgcn_spatial_wgrad_tile_slabs(2, 13, 25, 64, 64) == 2 * 3                # 6 frames per tile at V = 25: 3 tiles per sample
    assert lib.fgcn_spatial_bwd_tile_segments(2, 13, 25) == 3                           # 5 frames per tile there
    for bad in ((2, 13, 25, 96, 64), (2, 13, 25, 64, 32), (2, 13, 15, 64, 64), (2, 13, 33, 64, 64), (0, 13, 25, 64, 64)):
        assert lib.fgcn_spatial_wgrad_tile_slabs(*bad) == 0, bad
    assert lib.fgcn_spatial_bwd_tile_segments(2, 13, 15) == 0 and lib.fgcn_spatial_bwd_tile_segments(2, 0, 25) == 0
    try:
        assert lib.fgcn_set_tuning(16, 2) == 0 and lib.fgcn_spatial_wgrad_tile_slabs(B, 75, V, 256, 256) == 1
        assert lib.fgcn_set_tuning(15, 128) == 0 and lib.fgcn_spatial_bwd_tile_segments(B, 75, V) == 1
        assert lib.fgcn_set_tuning(31, 7) == 0
    finally:
        for k in (15, 16, 31):
            assert lib.fgcn_set_tuning(k, 0) == 0
    # the launcher validates on the host before any HIP call: null pointers, channel counts outside the 64s, a misaligned dY
    one = C.c_void_p(16)
    assert lib.fgcn_spatial_wgrad_tile(None, one, one, one, 2, 13, 25, 64, 64, 64, 64, 1, None) == -1
    assert lib.fgcn_set_math_mode(2) == 0
    try:
        assert lib.fgcn_spatial_wgrad_tile(one, one, one, one, 2, 13, 25, 96, 64, 96, 64, 1, None) == -1
        assert lib.fgcn_spatial_wgrad_tile(one, C.c_void_p(20), one, one, 2, 13, 25, 64, 64, 64, 64, 1, None) == -2
        assert b"aligned" in lib.fgcn_last_error()
        assert lib.fgcn_spatial_wgrad_tile(one, one, one, one, 2, 13, 25, 64, 64, 32, 64, 1, None) == -1     # ld_x < Cin
    finally:
        assert lib.fgcn_set_math_mode(0) == 0
    assert lib.fgcn_set_tuning(32, 1) == -1 and b"out of range" in lib.fgcn_last_error()
    assert lib.fgcn_set_tuning(-1, 1) == -1


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under fusion_gcn_amd/ may import it."""
    pkg = os.path.join(ROOT, "fusion_gcn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fusion_gcn_amd import ops
    x = torch.zeros(1, 2, 5, 4)
    with pytest.raises(_lib.FgcnError):
        ops.rows_gemm(x, torch.zeros(1, 4, 4), torch.zeros(1, 2, 5, 4), K=4, N=4)


def test_fresh_build_from_sources(tmp_path):
    """What a fresh clone does: every csrc/*.hip compiled from scratch for gfx950 (hipcc cross-compiles without a GPU) into an empty
    directory -- no prebuilt object or library of the tree is used -- and the result exports every symbol the header declares."""
    lib_path = build.build(out_dir=str(tmp_path))
    assert os.path.dirname(lib_path) == str(tmp_path) and os.path.getsize(lib_path) > 100_000
    objs = [f for f in os.listdir(tmp_path) if f.endswith(".o")]
    assert len(objs) == len(build.sources())
    import subprocess
    exported = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    for n in _declared_symbols():
        assert re.search(rf"\bT {n}\b", exported), f"fresh build does not export {n}"


def test_library_default_is_the_benchmarked_arithmetic(lib):
    """A drop-in that follows INTEGRATION.md section 2 alone (no set_math_mode call) computes in bf16x3 -- the float32-accurate
    arithmetic bench.py reports -- not in the 1.57x slower exact-f32 mode: checked in a fresh process, where no test
    fixture has selected a mode."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", "from fusion_gcn_amd import ops, _lib; print(ops.get_math_mode(), _lib.load().fgcn_get_products(), "
                        "_lib.load().fgcn_get_tuning(0), _lib.load().fgcn_get_tuning(1))"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["bf16x3", "0", "1", "0"], r.stdout


def test_every_autograd_function_of_the_package_is_context_bound(lib):
    """Every torch.autograd.Function the package defines runs its backward in its forward's library context (ops.context_bound);
    a module that defines Functions and forgets ops.bind_all_functions(globals()) would silently run them in the autograd
    thread's defaults.  A subclass of a bound Function is bound again (the marker is the class's OWN attribute)."""
    import importlib
    import pkgutil

    import torch

    import fusion_gcn_amd
    from fusion_gcn_amd import ops
    found = []
    for info in pkgutil.walk_packages(fusion_gcn_amd.__path__, "fusion_gcn_amd."):
        if info.name.endswith((".build", ".libfgcn")):        # (the build script; the C-ABI library is not a Python module)
            continue
        mod = importlib.import_module(info.name)
        for obj in vars(mod).values():
            if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function \
                    and obj.__module__ == mod.__name__:
                found.append(obj)
                assert "_fgcn_bound" in obj.__dict__, f"{obj.__module__}.{obj.__name__} is not context-bound"
    assert len(found) >= 10

    class Base(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            return g * 2

    ns = {"__name__": Base.__module__, "Base": Base}
    ops.bind_all_functions(ns)

    class Sub(Base):
        @staticmethod
        def forward(ctx, x):
            return x * 3
    ns["Sub"] = Sub
    assert "_fgcn_bound" not in Sub.__dict__
    x = torch.ones(2, requires_grad=True)
    Sub.apply(x).sum().backward()            # unbound subclass: the inherited backward finds no context and uses the defaults
    assert x.grad.tolist() == [2.0, 2.0]
    ops.bind_all_functions(ns)
    assert "_fgcn_bound" in Sub.__dict__


def test_host_geometry_queries_sweep(lib):
    """Every host-only query of the header (tile / slab / segment counts, availability) over a sweep of shapes, ragged ones and the
    limits included: non-negative, zero exactly where `*_available` says no, monotone in the sample count.  No launch is reached;
    test_host_code_under_sanitizers runs this sweep against the AddressSanitizer + UBSan build of the launchers' host code."""
    assert lib.fgcn_set_math_mode(2) == 0          # bf16x3: the mode the tile kernels exist in (availability is per mode, sizes are not)
    try:
        _geometry_sweep(lib)
    finally:
        assert lib.fgcn_set_math_mode(0) == 0


def _geometry_sweep(lib):
    for V in (15, 16, 18, 20, 22, 25, 27, 32, 33):
        for cin, cout in ((4, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (96, 64), (192, 192), (320, 256)):
            ic = cout // 4
            ok_f = lib.fgcn_spatial_fwd_tile_available(V, cin, cout)
            ok_b = lib.fgcn_spatial_bwd_tile_available(V, cin, cout)
            ok_w = lib.fgcn_spatial_wgrad_tile_available(V, cin, cout)
            ok_e = lib.fgcn_emb_tile_available(V, ic, cin)
            ok_ef = lib.fgcn_emb_fwd_tile_available(V, ic, cin)
            assert {ok_f, ok_b, ok_w, ok_e, ok_ef} <= {0, 1}
            prev = None
            for B, T in ((1, 1), (2, 13), (3, 75), (16, 150), (128, 300), (200, 301)):
                slabs = lib.fgcn_spatial_wgrad_tile_slabs(B, T, V, cin, cout)
                assert slabs >= 0 and (slabs > 0) == bool(ok_w), (B, T, V, cin, cout, slabs)
                eslabs = lib.fgcn_emb_wgrad_tile_slabs(B, T, V, ic, cin)
                assert eslabs >= 0 and (eslabs > 0) == bool(ok_e), (B, T, V, ic, cin, eslabs)
                if 16 <= V <= 32:
                    assert lib.fgcn_spatial_bwd_tile_segments(B, T, V) >= 1 and lib.fgcn_spatial_fwd_tile_tiles(B, T, V) >= B
                    assert lib.fgcn_emb_fwd_tile_segments(B, T, V, ic) >= (1 if ok_ef else 0)
                tiles = (lib.fgcn_tconv_halo_tiles(B, T, T, V), lib.fgcn_spatial_tiles(B, T), lib.fgcn_rows_gemm_tiles(B * T * V),
                         lib.fgcn_pw_gemm_tiles(B * T * V), lib.fgcn_elem_tiles(B * T * V), lib.fgcn_data_bn_tiles(B, T))
                assert all(t >= 1 for t in tiles), tiles
                if prev is not None:
                    assert all(a >= b for a, b in zip(tiles[:4], prev[:4])), (tiles, prev)      # more rows never need fewer tiles
                prev = tiles
    for n, nsplit in ((64, 1), (64, 8), (128, 4), (256, 16), (60, 3)):
        assert lib.fgcn_tconv_wgrad_slabs(n, nsplit) >= 1 and lib.fgcn_pw_wgrad_slabs(n, nsplit) >= 1


def test_host_code_under_sanitizers(lib):
    """the launchers' HOST code (argument validation, tile geometry, FastDiv tables, slab / segment counts, the
    per-thread contexts) built with -fsanitize=address,undefined (fusion_gcn_amd.build.build_host_asan; the gfx950 code objects are the
    usual ones -- GPU sanitizers do not exist on this pool) and driven by the host-only tests of this file and tests/test_fastdiv.py in
    a child interpreter that preloads the sanitizer runtime.  Clean = exit code 0 and no sanitizer report in the output."""
    import subprocess
    import sys
    asan_lib = build.build_host_asan()
    env = dict(os.environ, FGCN_LIB=asan_lib, LD_PRELOAD=build.asan_runtime(), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    keep = ("test_version_and_error_text or test_host_side_validation or test_tile_kernel_geometry_and_tuning_keys or "
            "test_host_geometry_queries_sweep or test_header_symbols_are_exported_and_bound or test_quotients or test_multiplier")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi.py"), os.path.join(ROOT, "tests", "test_fastdiv.py"),
                        "-q", "-x", "-p", "no:cacheprovider", "-k", keep], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    text = r.stdout + r.stderr
    assert r.returncode == 0, text[-4000:]
    assert "AddressSanitizer" not in text and "runtime error:" not in text, text[-4000:]
    assert "7 passed" in text or "passed" in text
