#!/usr/bin/env python3
"""Headline benchmark: clips/sec of AGCN forward+backward at (N,C,T,V,M) = (64,3,300,25,2) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one train-mode forward + backward (CrossEntropy, gradients for all 3.47 M parameters) of the
10-block AGCN (fusion_gcn_amd.models.mmargcn.agcn.Model, NTU-RGB-D graph, 60 classes) over synthetic clips already
resident in HBM.  Data parallel (`--gpus N`): ONE 64-clip batch is sharded over the N ranks (strong scaling, the north
star's "shard the clip batch across the 8 GPUs": 8 clips per GPU at N = 8, per-replica BatchNorm as in the reference's
DataParallel) and the step includes the single RCCL all-reduce of the flat 13.9 MB gradient buffer; the weak-scaling
reading (64 clips on every GPU) is measured right after it and reported as `other_scaling` (`--scaling weak` swaps the
two).  No optimizer step (the metric is fwd+bwd).  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` without a launcher starts the N ranks itself (a child `torch.distributed.run`, before this
process touches the GPU) and relays rank 0's JSON line.

Arithmetic (`--math`, reported in `dtype`): the default "bf16x3" forms every float32 product of the convolution / GEMM
kernels from exact three-way bfloat16 splits of both operands (six partial products, float32 accumulation: float32
accuracy -- the parity tests run it at the float32 tolerances) on the bf16 matrix pipe; "f32" issues
v_mfma_f32_32x32x2_f32 directly and is timed in the same run at N = 1 (`f32_mfma_mode`); "bf16" is BASELINE config 5.

Extra objects in that line:
  roofline      the dominant kernel (halo-tile 9x1 temporal conv), timed live with HIP events on the stream it runs on,
                against the matrix peak of its arithmetic (MI355X_MICROARCH.md): bf16 dense 2500 TFLOP/s / 6 partial
                products = 416.7 TFLOP/s of float32-equivalent work in bf16x3 mode, 157.3 TFLOP/s for the f32 MFMA.
  cpu_baseline  the CPU oracle (oracle/agcn_oracle.py = stock-torch restatement of the reference model) timed on
                this box's host cores on a bounded sample of the same workload (N = 16 clips).
  parity_at_full_shape  the oracle outputs of that sample are kept: the HIP model with the same state on the same 16 full-size
                clips -- logits, loss, flat gradient with ReLU-flip accounting (oracle/relu_masks.py).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA (32x32x16 / 16x16x32), dense
PEAK_SPLIT3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6   # float32-equivalent FLOPs when every product costs six bf16 MFMAs
PEAK_SPLIT2H_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3  # f16 MFMAs run at the bf16 rate; three products per product group
MATH_PEAK = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16x3": PEAK_SPLIT3_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "f16x2": PEAK_SPLIT2H_TFLOPS}
MATH_DTYPE = {
    "f32": "f32",
    "bf16x3": "f32 (products from exact 3-way bf16 splits of both operands: 6 bf16 MFMAs per product group, f32 accumulate; "
              "f32 storage / statistics; same parity tolerances as the f32 MFMA path)",
    "bf16": "bf16 MFMA operands and bf16 storage of the activation-sized tensors (the reference's autocast semantics; paths.half_activations), "
            "f32 accumulate / BatchNorm statistics / softmax",
    "f16x2": "f32 (bf16x3, with the temporal and 1x1 convolutions' products from block-scaled 2-way f16 splits of both operands: 3 f16 "
             "MFMAs per product group, f32 accumulate; f32 storage / statistics; same parity tolerances as the f32 MFMA path)"}
MATH_DTYPE_SHORT = {"f32": "f32", "bf16x3": "f32 (products from exact 3-way bf16 splits; see notes.dtype)", "bf16": "bf16 operands and activation storage, f32 accumulate",
                    "f16x2": "f32 (products from block-scaled 2-way f16 splits; see notes.dtype)"}
MATH_KERNEL = {"f32": "conv_halo_kernel<{nt},3>", "bf16": "conv_halo_x3k32_kernel<{nt2},32,1> (one bf16 part)",
               "bf16x3": "conv_halo_x3k32_kernel<{nt2},32>", "f16x2": "conv_halo_x3k32_kernel<{nt2},32,2> (f16x2 products)"}
PEAK_HBM_GBPS = 8000.0            # spec; ~6300 achievable
ACHIEVABLE_HBM_GBPS = 6300.0
SHAPE = dict(N=64, M=2, T=300, V=25, C=3, classes=60)


def block_table():
    plan, cin = [], SHAPE["C"]
    for i, cout in enumerate([64] * 4 + [128] * 3 + [256] * 3):
        plan.append((cin, cout, 2 if i in (4, 7) else 1, i != 0))
        cin = cout
    return plan


def algorithmic_costs(n_clips: int):
    """FLOPs and HBM bytes of fwd+bwd for n_clips clips (SURVEY.md §8d formulas; bwd = 2x fwd)."""
    B, T, V = n_clips * SHAPE["M"], SHAPE["T"], SHAPE["V"]
    flops = byts = 0.0
    for cin, cout, s, res in block_table():
        ic, Tp = cout // 4, (T - 1) // s + 1
        down = cin != cout
        conv_res = res and (down or s != 1)
        f = B * T * V * (12 * cin * ic + 6 * V * cin + 6 * cin * cout + (2 * cin * cout if down else 0))
        f += 6 * V * V * ic * T * B
        f += B * Tp * V * (18 * cout * cout + (2 * cin * cout if conv_res else 0))
        flops += 3 * f
        byts += 4 * B * V * (3 * cin * T + 5 * cout * T + 2 * cout * Tp)
        T = Tp
    return flops, byts


def build_model(device):
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from fusion_gcn_amd.models.mmargcn.agcn import Model
    from fusion_gcn_amd.util import Graph
    torch.manual_seed(1)
    if SHAPE["V"] == 25:
        model = Model((SHAPE["M"], SHAPE["T"], SHAPE["V"], SHAPE["C"]), SHAPE["classes"],
                      Graph(ntu.skeleton_edges, center_joint=ntu.center_joint))
    else:   # BASELINE configs 3 / 4: IMU modalities as extra joints (mode skeleton_imu_spatial_fusion)
        from fusion_gcn_amd.datasets.mmact import constants as mmact
        from fusion_gcn_amd.models.mmargcn.mmargcn import Model as MM
        c, n_imu = (ntu, 2) if SHAPE["V"] == 27 else (mmact, 4)
        model = MM({"skeleton": (SHAPE["M"], SHAPE["T"], SHAPE["V"], SHAPE["C"])}, SHAPE["classes"],
                   Graph(c.skeleton_edges, center_joint=c.center_joint), mode="skeleton_imu_spatial_fusion",
                   num_imu_joints=n_imu, imu_enhanced_mode="append_center")
    # the reference initialises the gcn BatchNorm scale and adj_b at 1e-6 (blocks start as near-identities); use O(1)
    # values so every kernel sees realistic magnitudes (timing does not depend on them, ReLU sparsity does slightly)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gcn1.bn.weight"):
                p.fill_(1.0)
    return model.to(device).train()


def time_dominant_kernel(device, b_local: int, reps: int = 10, widths=(64, 128, 256)):
    """Live HIP-event timing of the halo-tile temporal-conv kernel (conv_halo_kernel: with rows_wgrad_kernel the
    largest share of the step) at the three channel widths of the model, forward form with bias and BatchNorm
    partial sums exactly as the block launches it; returns per-launch algorithmic FLOPs and mean duration."""
    from fusion_gcn_amd import ops
    out = []
    T = SHAPE["T"]
    for c, t in ((64, T), (128, (T - 1) // 2 + 1), (256, ((T - 1) // 2) // 2 + 1)):
        if c not in widths:
            continue
        x = torch.randn(b_local, t, SHAPE["V"], c, device=device)
        w4 = ops.pack_conv(torch.randn(9, c, c, device=device) * (9 * c) ** -0.5)   # the current math mode's form
        bias = torch.randn(c, device=device)
        if ops.get_math_mode() == "bf16" and ops.paths().get("half_activations", "bf16"):
            x = x.to(torch.bfloat16)        # the form the step launches in this mode: bfloat16 G in, bfloat16 U out (fgcn_tconv_halo_t)
        y = torch.empty_like(x)

        def launch():
            ops.tconv_halo(x, w4, y, Th=t, taps=9, tb=1, tc=-4, bias=bias, stats=True)
        for _ in range(2):
            launch()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()                      # our launches go to torch's current stream: the events see them
        for _ in range(reps):
            launch()
        end.record()
        end.synchronize()
        ms = start.elapsed_time(end) / reps
        flops = 2.0 * b_local * t * SHAPE["V"] * 9 * c * c
        byts = float(x.element_size()) * b_local * t * SHAPE["V"] * 2 * c
        out.append(dict(channels=c, frames=t, ms=ms, flops=flops, bytes=byts, tflops=flops / ms / 1e9))
    return out


def cpu_baseline(n_clips: int = 16, iters: int = 2, device=None, maths=("bf16x3",)):
    """Reported baseline, not the target: the oracle model, fwd+bwd, on the host cores of this box.
    torch's CPU convolutions stop scaling (and then collapse) well below the 256 hardware threads of the GPU box,
    so the thread count is calibrated on one clip first and the best one is used and reported as ``cores``.
    -> (cpu_baseline dict, {math mode: parity_at_full_shape dict} or None).  With ``device`` the oracle's results on those n_clips clips of the
    FULL (C,T,V,M) shape are not thrown away: the HIP model is loaded with the same state and run on the same clips, and logits,
    loss and the flat gradient (with ReLU-flip accounting, oracle/relu_masks.py) are compared -- the oracle as the checker."""
    from fusion_gcn_amd.datasets.ntu_rgb_d import constants as ntu
    from oracle import agcn_oracle as O
    from oracle import filler, graph_oracle
    from oracle import relu_masks as RM
    hw = os.cpu_count() or 1
    adj = graph_oracle.spatial_partition_stack(ntu.skeleton_edges)
    sd = O.new_state_dict((SHAPE["M"], SHAPE["T"], SHAPE["V"], SHAPE["C"]), SHAPE["classes"], adj)
    for k in list(sd):
        if not k.endswith("adj_a"):
            sd[k] = torch.from_numpy(filler.fill_value_for(k, tuple(sd[k].shape))).reshape(sd[k].shape).to(sd[k].dtype)
        if k.endswith("gcn1.bn.weight"):
            sd[k] = torch.ones_like(sd[k])           # as build_model (config.init_override)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n_clips, SHAPE["M"], SHAPE["T"], SHAPE["V"], SHAPE["C"], generator=g)
    y = torch.randint(0, SHAPE["classes"], (n_clips,), generator=g)
    best = (float("inf"), 1)
    for thr in sorted({t for t in (8, 16, 32) if t <= hw} or {hw}):
        torch.set_num_threads(thr)
        O.loss_and_grads(x[:1], y[:1], sd)  # warm-up at this thread count
        t0 = time.perf_counter()
        O.loss_and_grads(x[:1], y[:1], sd)
        best = min(best, (time.perf_counter() - t0, thr))
    cores = best[1]
    torch.set_num_threads(cores)
    hip = None
    if device is not None and SHAPE["V"] == 25:
        from fusion_gcn_amd.models.mmargcn.agcn import Model
        from fusion_gcn_amd.util import Graph
        hip = Model((SHAPE["M"], SHAPE["T"], SHAPE["V"], SHAPE["C"]), SHAPE["classes"], Graph(ntu.skeleton_edges, center_joint=ntu.center_joint))
        hip.load_state_dict(sd, strict=True)
    if hip is not None:                     # the warm-up run is the one whose outputs are kept for the comparison
        oracle = RM.oracle_side(x, y, sd, [k for k, _ in hip.named_parameters()])
    else:
        O.loss_and_grads(x, y, sd)          # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        O.loss_and_grads(x, y, sd)
    dt = (time.perf_counter() - t0) / iters
    base = dict(value=round(n_clips / dt, 3), unit="clips/s", cores=cores, kind="port",
                sample=f"oracle (stock-torch restatement of the reference model) fwd+bwd, {n_clips} clips of the same "
                       f"(C,T,V,M)=(3,300,25,2) workload, 1 warm-up + {iters} timed iterations, {dt:.2f} s/iter, "
                       f"{cores} torch threads (best of 8/16/32 on a {hw}-thread host)")
    parity = None
    if hip is not None:
        import math as _m
        from fusion_gcn_amd import ops as _ops
        hip = hip.to(device).train()
        parity = {}
        for math in maths:
            with _ops.math_mode(math):
                rep = RM.gradient_parity_report(hip, x.to(device), y.to(device), oracle=oracle)
            bound = 1e-4 + 2.0 * _m.sqrt(rep["flips"] / (rep["decisions"] / 20))
            parity[math] = {"clips": n_clips, "shape_CTVM": [SHAPE["C"], SHAPE["T"], SHAPE["V"], SHAPE["M"]], "math": math,
                            "against": "float32 CPU oracle run of cpu_baseline (same state, same clips)",
                            "logits_rel_l2": float(f"{rep['logits_err']:.3e}"), "loss_abs_err": float(f"{rep['loss_err']:.3e}"),
                            "relu_flips": rep["flips"], "relu_decisions": rep["decisions"],
                            "flat_grad_rel_l2_as_is": float(f"{rep['err_plain']:.3e}"),
                            "flat_grad_rel_l2_with_oracle_relu_decisions": float(f"{rep['err_injected']:.3e}"),
                            "ok": bool(rep["logits_err"] < 1e-5 and rep["loss_err"] < 1e-5 and rep["err_injected"] < 1e-4
                                       and rep["err_plain"] <= bound)}
        del hip
    return base, parity


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


TRAFFIC_RECORDS = ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")      # newest first


def measured_traffic(dom, samples, math="f32"):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, gfx950 correction of MI355X_MICROARCH.md applied: FETCH_SIZE counts 16-byte-per-lane
    reads at half their bytes).  A STORED figure (PMC counters cannot be read from inside the benchmark process): only valid
    for the kernel and shape it was collected at; None otherwise.  `tools/collect_profiles.sh` re-collects it."""
    rec = None
    for name in TRAFFIC_RECORDS:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)[{"f32": "conv_halo_fwd", "bf16x3": "conv_halo_x3_fwd", "f16x2": "conv_halo_f16x2_fwd"}[math]]
            break
        except (OSError, KeyError, ValueError):
            continue
    if rec is None:
        return None
    if rec["channels"] != dom["channels"] or rec["frames"] != dom["frames"] or rec["samples"] != samples:
        return None
    return rec["traffic_bytes"]


def live_traffic(math: str, batch: int, timeout_s: int = 90):
    """HBM-side bytes per launch of the dominant kernel MEASURED in this run: two child processes, `rocprofv3 --kernel-trace --pmc FETCH_SIZE`
    and `--pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes; nothing but --kernel-trace beside --pmc), each around
    `python3 bench.py --kernel-only` (the same launches `time_dominant_kernel` times).  traffic = 2 x FETCH_SIZE + WRITE_SIZE (the guide's
    gfx950 correction: 16-byte-per-lane reads are tallied at half their bytes), KiB -> bytes, averaged over the kernel's launches.
    -> (bytes or None, how it went).  Any failure (no rocprofv3, a timeout, an unreadable file) returns None: the caller then reports the
    stored record and says so."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="fgcn_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--kernel-only", "--math", math, "--batch", str(batch)]
        env = dict(os.environ, TMPDIR="/tmp")
        env.pop("WORLD_SIZE", None)
        # the child is a process GROUP (rocprofv3 -> python -> the kernels): on a timeout the whole group is killed, so that no
        # grandchild keeps the GPU busy under the timed runs that follow, and the scratch directory goes either way
        proc = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.wait()
                return None, f"rocprofv3 --pmc {counter} timed out after {timeout_s} s (process group killed)"
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            vals = []
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and "conv_halo" in row.get("Kernel_Name", ""):
                            vals.append(float(row["Counter_Value"]))
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
        if rc != 0 or not vals:
            return None, f"rocprofv3 --pmc {counter}: rc {rc}, {len(vals)} launches read"
        got[counter] = (sum(vals) / len(vals), len(vals))
    byts = int(round((2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024))
    return byts, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two child passes of `bench.py --kernel-only`, "
                  f"{got['FETCH_SIZE'][1]} / {got['WRITE_SIZE'][1]} launches of the kernel), 2*FETCH_SIZE + WRITE_SIZE per launch")


def metric_name(n_global: int) -> str:
    return "clips/sec (N,C,T,V,M)=(%d,3,300,%d,2) fwd+bwd" % (n_global, SHAPE["V"])


def collective_report(world: int, rank: int, backend: str, device=None, numel: int = 3_469_510, reps: int = 20):
    """What the N > 1 line carries so that it proves which collective ran where: backend name, world size, every rank's local device
    index and device name (gathered), the library version (RCCL's, as torch reports it for the "nccl" backend), and the time of ONE
    all-reduce of a buffer of the gradient exchange's size, timed alone (barrier + synchronize on both sides, max over ranks, `reps`
    repetitions).  Collective calls: every rank must call this.  -> dict on every rank (rank 0 prints it)."""
    on_gpu = device is not None and device.type == "cuda"
    if on_gpu:
        me = {"rank": rank, "local_device": device.index, "name": torch.cuda.get_device_name(device),
              "uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", ""))}
    else:
        me = {"rank": rank, "local_device": None, "name": "cpu", "uuid": ""}
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    buf = torch.ones(numel, device=device if on_gpu else "cpu", dtype=torch.float32)

    def fence():
        dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()
    for _ in range(3):
        dist.all_reduce(buf)
    fence()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf)
    fence()
    t = torch.tensor([(time.perf_counter() - t0) / reps], device=device if on_gpu else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001 - the version string is a label, not a result
            version = f"unavailable ({type(e).__name__})"
    return {"backend": backend, "library": "RCCL (torch.distributed backend \"nccl\" on ROCm)" if backend == "nccl" else backend,
            "rccl_version": version, "world": world, "devices": everyone, "distinct_devices": len({(d["name"], d["local_device"], d["uuid"]) for d in everyone}),
            "allreduce_bytes": numel * 4, "allreduce_ms": round(1e3 * float(t), 4), "allreduce_reps": reps}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: run the same command line under torch.distributed.run (one rank
    per GPU, rendezvous on 127.0.0.1) as a CHILD process and pass its output and exit code through.  This process never
    initialises the GPU, so nothing is re-exec'ed after a HIP call."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"starting {n} ranks: {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world: int, rank: int) -> None:
    """The multi-rank control flow of the benchmark without a GPU (tests/test_bench_host.py): rendezvous, the shard of the
    global batch this rank would hold, one all-reduce of a buffer of the gradient exchange's size, max-over-ranks timing,
    rank 0's JSON line.  Not a measurement: value is null."""
    from fusion_gcn_amd.dp import shard_batch
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("FGCN_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
    n_global = args.batch * world if args.scaling == "weak" else args.batch
    shard = shard_batch(n_global, rank, world)
    flat = torch.full((3_469_510,), float(rank + 1))
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
        dist.all_reduce(flat)
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(flat[0]) == world * (world + 1) / 2
    coll = collective_report(world, rank, os.environ.get("FGCN_BENCH_BACKEND", "gloo"), reps=3) if world > 1 else None
    if rank == 0:
        line = {"metric": metric_name(n_global), "value": None, "unit": "clips/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                "scaling": args.scaling, "vs_baseline": None, "dry_run": True, "data": "synthetic",
                "config": {"global_batch": n_global, "per_gpu_batch": shard.stop - shard.start,
                           "parallelism": f"dp{world}", "exchange_s": round(float(t), 4)}}
        if coll:
            line["collective"] = coll
        if world > 1 and not args.no_cpu_baseline:      # the N > 1 line carries the CPU baseline too (rank 0 times it, the others wait)
            line["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": None, "kind": "port", "sample": "dry run: not timed"}
        print(json.dumps(line), flush=True)
    if world > 1:
        wait_for_rank0(rank)      # the same hand-off as the measured run's (there rank 0 times the CPU oracle first)
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=SHAPE["N"], help="clips per GPU (weak) / global clip batch (strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default): ONE batch of --batch clips sharded over the GPUs; weak: --batch clips on every GPU")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check without a GPU: rendezvous, batch sharding, one all-reduce of a gradient-sized buffer "
                         "over FGCN_BENCH_BACKEND (gloo), then the JSON line with value null")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--math", choices=("f32", "bf16", "bf16x3", "f16x2"), default="bf16x3",
                    help="bf16x3 (default): f32-accurate split-bf16 products; f32: v_mfma_f32 directly; bf16: BASELINE "
                         "config 5 (bf16 MFMA operands, f32 accumulation)")
    ap.add_argument("--no-f32-mode", action="store_true",
                    help="N = 1 only: skip the secondary timing of the exact-f32-MFMA mode (f32_mfma_mode)")
    ap.add_argument("--no-bf16-mode", action="store_true",
                    help="N = 1 only: skip the secondary timing of BASELINE config 5 (math mode bf16: bf16 operands and activation storage; modes.bf16_config5)")
    ap.add_argument("--no-f16x2-mode", action="store_true",
                    help="N = 1 only: skip the secondary timing of the f16x2 products (f16x2_mode)")
    ap.add_argument("--verify-dp", action="store_true",
                    help="N > 1 only: check the exchanged gradient buffer of the (graph) step against an eager step")
    ap.add_argument("--no-other-scaling", action="store_true",
                    help="N > 1 only: skip the secondary (strong-scaling) measurement reported as other_scaling")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--allow-eager", action="store_true",
                    help="if the HIP-graph capture of the step fails, time the eagerly launched step instead of exiting non-zero")
    ap.add_argument("--joints", type=int, choices=(25, 27, 22), default=25,
                    help="25: the headline (BASELINE config 2); 27: config 3 (NTU graph + 2 IMU joints); 22: config 4 (MMAct COCO-18 + "
                         "4 IMU joints, 35 classes) -- parity-test shapes timed for the record, not the headline")
    ap.add_argument("--loader", choices=("none", "resident", "streaming"), default="none",
                    help="feed every timed step from a feature file in the reference's on-disk format through "
                         "fusion_gcn_amd.data.ClipBatches (resident: split uploaded once, device-side gather; streaming: pinned "
                         "double-buffered H2D copies on a side stream = the PCIe-inclusive rate); default: one HBM-resident batch")
    ap.add_argument("--optimizer", choices=("none", "adam", "sgd"), default="none",
                    help="also run the fused parameter update (fusion_gcn_amd.optim.FlatOptimizer: ADAM weight_decay 0.01 as "
                         "config/utd-mhad/skeleton/agcn.yaml, or SGD momentum 0.9 nesterov) inside every timed step; the "
                         "headline metric is fwd+bwd, so the default leaves it out")
    ap.add_argument("--paths", default="",
                    help="A/B only: kernel-form options of this run's library context, 'name=value,name=value' (fusion_gcn_amd/paths.py: "
                         "emb_tile=0, spatial_tile_min_cout=64, bn_sums_in_dgrad=0, fuse_g=1, ...); the same string as the FGCN_PATHS "
                         "environment variable, which sets the process defaults")
    ap.add_argument("--keep-packed", action="store_true",
                    help="A/B only (not the headline): keep the packed / split weight forms across steps instead of "
                         "rebuilding them from the parameters inside every timed step")
    ap.add_argument("--tune", default="", help="fgcn_set_tuning pairs for A/B runs, e.g. 6=21505")
    ap.add_argument("--no-inference", action="store_true",
                    help="N = 1 only: skip the secondary timing of the inference forward (eval mode, no autograd: the fused output stages)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1 only: report the stored PMC record as roofline.traffic instead of measuring it in two rocprofv3 child passes")
    ap.add_argument("--kernel-only", action="store_true",
                    help="only the live timing of the dominant kernel at its dominant shape (256 channels): the command "
                         "profiles/*_dominant_kernel_stats.csv is the rocprofv3 --kernel-trace --stats summary of")
    args = ap.parse_args()
    if args.joints != 25:
        SHAPE["V"] = args.joints
        SHAPE["classes"] = 35 if args.joints == 22 else 60

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))     # nothing has touched the GPU yet: the ranks are fresh child processes
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.batch % world:
        raise SystemExit(f"--batch {args.batch} is not divisible by {world} ranks")
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    # FGCN_BENCH_BACKEND=gloo lets the multi-rank control flow be exercised on a box with fewer GPUs than ranks (ranks
    # share devices, collectives go through the host): a plumbing check, never a measurement.
    backend = os.environ.get("FGCN_BENCH_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (generous collective timeout: graph capture + warm-up of one rank may lag the others by minutes on a cold box; the wait for
        # rank 0's CPU baseline at the end does NOT sit in a collective at all -- see wait_for_rank0 below)
        import datetime
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=datetime.timedelta(minutes=60))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=60))

    from fusion_gcn_amd import ops as _ops
    _ops.set_math_mode(args.math)
    for kv in filter(None, args.tune.split(",")):
        from fusion_gcn_amd import _lib as _flib
        _flib.load().fgcn_set_tuning(*(int(t) for t in kv.split("=")))
    if args.kernel_only:
        kern = time_dominant_kernel(device, args.batch * SHAPE["M"], reps=20, widths=(256,))
        print(json.dumps({"kernel": MATH_KERNEL[args.math].format(nt=4, nt2=2) + " forward, 256 channels", **kern[0]}), flush=True)
        return
    if args.paths:
        _ops.paths().update_from(args.paths)      # this thread's current context (the process defaults here): per context, not a module global
        log(f"path options: {args.paths}")
    from fusion_gcn_amd.dp import FlatGradients, broadcast_parameters, shard_batch
    from fusion_gcn_amd.loss import cross_entropy      # nn.CrossEntropyLoss() of the reference's session, on libfgcn
    model = build_model(device)
    broadcast_parameters(model)
    grads = FlatGradients(model.parameters())
    opt = None
    if args.optimizer != "none":       # before any graph capture: the parameters move into one flat buffer
        from fusion_gcn_amd.optim import FlatOptimizer
        opt = (FlatOptimizer(model.parameters(), "ADAM", 1e-5, grads=grads, weight_decay=0.01) if args.optimizer == "adam"
               else FlatOptimizer(model.parameters(), "SGD", 1e-5, grads=grads, momentum=0.9, nesterov=True, weight_decay=1e-4))

    n_global = args.batch * world if args.scaling == "weak" else args.batch

    def resident_shard(n_total):
        """This rank's clips of a seeded synthetic batch of n_total clips, resident in HBM before any timed region."""
        shard = shard_batch(n_total, rank, world)
        g = torch.Generator().manual_seed(1)
        x_all = torch.randn(n_total, SHAPE["M"], SHAPE["T"], SHAPE["V"], SHAPE["C"], generator=g)
        y_all = torch.randint(0, SHAPE["classes"], (n_total,), generator=g)
        return x_all[shard].to(device).contiguous(), y_all[shard].to(device), shard.stop - shard.start

    def make_step(x, y):
        """-> (step function, launch mode).  The step is ~630 kernel launches; at 8 clips per GPU their host cost exceeds
        the GPU time, so forward + backward are captured once into a HIP graph (our ctypes launches go to torch's capturing
        stream) and replayed; inputs, parameters and gradient buffers are static, the data-parallel exchange stays
        outside the graph."""
        def fwd_bwd():
            # a training step follows an optimizer update, so the packed / split weight forms the kernels stream are rebuilt
            # from the parameters inside every timed step (the blocks cache them per parameter version otherwise)
            for m in model.modules():
                if hasattr(m, "mark_packed_stale") and not args.keep_packed:
                    m.mark_packed_stale()
            loss = cross_entropy(model(x), y)
            loss.backward()
            return loss

        def fwd_bwd_checked():
            grads.zero()
            loss = fwd_bwd()
            torch.cuda.synchronize()
            return loss.detach()

        def step_eager():
            grads.zero()
            loss = fwd_bwd()
            if world > 1:
                grads.all_reduce_mean()     # gather into the flat buffer + ONE RCCL all-reduce + 1/world
            if opt is not None:
                opt.step()                  # one launch over the flat parameter / gradient / state buffers
            return loss

        if args.no_graph:
            return step_eager, "eager"
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    step_eager()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            eager_loss = float(fwd_bwd_checked())
            graph = torch.cuda.CUDAGraph()
            grads.zero()
            with torch.cuda.graph(graph):
                static_loss = fwd_bwd()
                if world > 1 or opt is not None:
                    grads.gather()      # the copy into the flat exchange buffer is part of the replayed step
            # a replayed graph must reproduce the eager loss (parameters do not change): twice, with a sync in between
            for _ in range(2):
                graph.replay()
                torch.cuda.synchronize()
                got = float(static_loss.detach())
                if abs(got - eager_loss) > 1e-4 * max(1.0, abs(eager_loss)):
                    raise RuntimeError(f"graph replay loss {got} != eager loss {eager_loss}")

            def step_graph():
                graph.replay()
                if world > 1:
                    grads.all_reduce_mean()
                if opt is not None:
                    opt.step()
                return static_loss
            return step_graph, "hipgraph"
        except Exception as e:  # noqa: BLE001 - a bench that silently changes its launch mode measures something else
            log(f"graph capture failed ({type(e).__name__}: {e})")
            if not args.allow_eager:
                raise SystemExit("bench.py: HIP-graph capture of the step failed (see the message above); --allow-eager runs the "
                                 "eagerly launched step instead, --no-graph asks for it from the start")
            torch.cuda.synchronize()
            return step_eager, "eager"

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        """Seconds for `steps` steps after `warmup` untimed ones: barrier + synchronize on both sides, max over ranks."""
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t)
        return elapsed, float(loss.detach())

    x, y, n_local = resident_shard(n_global)
    step, mode = make_step(x, y)
    pipeline = None
    if args.loader != "none":
        # a synthetic split in the reference's file format (4 global batches of clips), read back through the dataset interface
        import tempfile

        import numpy as np

        from fusion_gcn_amd.data import ClipBatches, MultiModalDataset, NumpyDatasetLoader, NumpyWriter
        root = tempfile.mkdtemp(prefix=f"fgcn_bench_r{rank}_")
        n_file = 4 * n_global
        rng = np.random.default_rng(1)
        with NumpyWriter(os.path.join(root, "skeleton_train_features.npy"), np.float32,
                         (n_file, SHAPE["M"], SHAPE["T"], SHAPE["V"], SHAPE["C"])) as w:
            for _ in range(n_file):
                w.collect_next(rng.standard_normal((SHAPE["M"], SHAPE["T"], SHAPE["V"], SHAPE["C"]), dtype=np.float32))
        np.save(os.path.join(root, "train_labels.npy"), rng.integers(0, SHAPE["classes"], n_file))
        batches = ClipBatches(MultiModalDataset([(root, NumpyDatasetLoader())], "train"), n_global, shuffle=True, drop_last=True,
                              seed=1, rank=rank, world=world, device=device, resident=args.loader == "resident")
        compute_step = step

        def epochs():
            e = 0
            while True:
                batches.set_epoch(e)
                yield from batches
                e += 1
        feed = epochs()

        def step():         # noqa: F811 - the timed step now starts with the batch hand-over into the step's static input
            feats, lab, _ = next(feed)
            x.copy_(feats, non_blocking=True)
            y.copy_(lab, non_blocking=True)
            return compute_step()
        pipeline = {"mode": args.loader, "file_clips": n_file, "bytes_per_step_h2d": 0 if args.loader == "resident"
                    else int(n_local * SHAPE["M"] * SHAPE["T"] * SHAPE["V"] * SHAPE["C"] * 4)}
    log(f"model + {n_local} clips resident on {device}; warm-up x{args.warmup}, timing {args.steps} steps")
    elapsed, loss_val = timed(step, args.steps, args.warmup)
    log(f"{args.steps} steps in {elapsed:.3f} s")

    # With more than one GPU the other reading of "data-parallel scaling" is reported beside the headline number in the
    # same line: ONE batch of --batch clips sharded over the ranks (strong scaling; 8 clips per GPU at N = 8).
    other = None
    if world > 1 and not args.no_other_scaling and args.batch >= world:
        # (no try/except: a rank that skipped a collective would leave the others hanging; an error ends the job loudly)
        n_other = args.batch if args.scaling == "weak" else args.batch * world
        x2, y2, n_local2 = resident_shard(n_other)
        step2, mode2 = make_step(x2, y2)
        steps2 = max(args.steps, 10)
        el2, _ = timed(step2, steps2, max(args.warmup, 2))
        other = {"scaling": "strong" if args.scaling == "weak" else "weak", "global_batch": n_other,
                 "per_gpu_batch": n_local2, "value": round(n_other * steps2 / el2, 2), "unit": "clips/s",
                 "ms_per_step": round(1e3 * el2 / steps2, 3), "steps": steps2, "launch": mode2}
        del step2, x2, y2

    if args.verify_dp and world > 1:   # after all timed regions
        # the replayed step must leave the same averaged gradients in the flat buffer as an eager step
        step()
        torch.cuda.synchronize()
        got = grads.flat.clone()
        grads.zero()
        cross_entropy(model(x), y).backward()
        grads.all_reduce_mean()
        torch.cuda.synchronize()
        err = float((got - grads.flat).norm() / grads.flat.norm())
        log(f"verify-dp: flat gradient buffer, {mode} step vs eager step: rel-L2 {err:.2e}")
        if err != 0.0:      # data_bn and the loss run on libfgcn too: every gradient of the step is a fixed-order sum
            raise SystemExit("verify-dp failed: the replayed step's gradient buffer must equal the eager step's bit for bit")

    coll = collective_report(world, rank, backend, device=device, numel=int(grads.flat.numel())) if world > 1 else None
    kern = None if args.no_kernel_timing else time_dominant_kernel(device, n_local * SHAPE["M"])
    f32_mode = None
    if world == 1 and args.math == "bf16x3" and not args.no_f32_mode:
        # the same step with v_mfma_f32_32x32x2_f32 issued directly, for reference beside the headline number
        _ops.set_math_mode("f32")
        step_f, mode_f = make_step(x, y)
        el_f, loss_f = timed(step_f, args.steps, args.warmup)
        f32_mode = {"value": round(n_global * args.steps / el_f, 2), "unit": "clips/s",
                    "ms_per_step": round(1e3 * el_f / args.steps, 3), "loss": round(float(loss_f), 5), "launch": mode_f}
        _ops.set_math_mode(args.math)
        del step_f
    f16x2_mode = None
    if world == 1 and args.math == "bf16x3" and not args.no_f16x2_mode:
        # the same step with the temporal / 1x1 convolutions and their weight gradients on block-scaled two-way f16 splits (three
        # MFMAs per product group instead of six; float32-class: tests/test_f16x2_edge_gpu.py) -- reported beside
        # the headline, which stays on the arithmetic the earlier rounds were judged on
        _ops.set_math_mode("f16x2")
        step_h, mode_h = make_step(x, y)
        el_h, loss_h = timed(step_h, args.steps, args.warmup)
        kern_h = None if args.no_kernel_timing else time_dominant_kernel(device, n_local * SHAPE["M"], widths=(256,))
        f16x2_mode = {"value": round(n_global * args.steps / el_h, 2), "unit": "clips/s",
                      "ms_per_step": round(1e3 * el_h / args.steps, 3), "loss": round(float(loss_h), 5), "launch": mode_h,
                      "dtype": MATH_DTYPE["f16x2"]}
        if kern_h:
            f16x2_mode["dominant_kernel"] = {"kernel": MATH_KERNEL["f16x2"].format(nt2=2) + " (9x1 temporal conv forward, 256 channels)",
                                             "ms_per_launch": round(kern_h[0]["ms"], 4), "achieved_tflops": round(kern_h[0]["tflops"], 2),
                                             "peak_tflops": round(PEAK_SPLIT2H_TFLOPS, 1),
                                             "frac": round(kern_h[0]["tflops"] / PEAK_SPLIT2H_TFLOPS, 4)}
        _ops.set_math_mode(args.math)
        del step_h
    bf16_mode = None
    if world == 1 and args.math == "bf16x3" and not args.no_bf16_mode and SHAPE["V"] == 25:
        # BASELINE config 5 on the same model and batch: bf16 MFMA operands AND bf16 storage of the activation-sized tensors (the reference's
        # autocast semantics, DESIGN.md section 3.14 b / g) -- a different numerical contract, reported beside the headline
        _ops.set_math_mode("bf16")
        step_b, mode_b = make_step(x, y)
        el_b, loss_b = timed(step_b, args.steps, args.warmup)
        bf16_mode = {"value": round(n_global * args.steps / el_b, 2), "unit": "clips/s", "ms_per_step": round(1e3 * el_b / args.steps, 3),
                     "loss": round(float(loss_b), 5), "launch": mode_b, "dtype": MATH_DTYPE["bf16"]}
        _ops.set_math_mode(args.math)
        del step_b
    inference = None
    if world == 1 and not args.no_inference and args.math in ("bf16x3", "bf16") and args.loader == "none":
        # the same model's INFERENCE forward (eval-mode BatchNorm, no autograd graph): BatchNorm + shortcut + ReLU are then the epilogues of
        # the two north-star kernels (fgcn_spatial_fwd_tile_bn_relu, fgcn_tconv_halo_bn_relu) -- reported beside the training step
        try:
            model.eval()
            with torch.no_grad():
                for _ in range(2):
                    model(x)
                torch.cuda.synchronize()
                ig = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ig):
                    model(x)
                ig.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    ig.replay()
                torch.cuda.synchronize()
                dt_i = (time.perf_counter() - t0) / args.steps
            inference = {"value": round(n_global / dt_i, 1), "unit": "clips/s", "ms_per_batch": round(1e3 * dt_i, 3),
                         "what": "forward only, eval-mode BatchNorm, torch.no_grad(), HIP-graph replay; BatchNorm + shortcut + ReLU in the "
                                 "epilogues of the spatial and temporal kernels (paths.fused_inference)"}
            del ig
        except Exception as e:  # noqa: BLE001 - a side number must not cost the headline line
            log(f"inference timing failed ({type(e).__name__}: {e})")
        finally:
            model.train()
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        clips_per_s = n_global * args.steps / elapsed
        flops, byts = algorithmic_costs(n_global)
        # Key order: a reader (or a record that keeps only the head of the line) gets every NUMBER first -- headline, the two side
        # modes, whole-step fractions, roofline, CPU baseline -- then the short descriptors, and the long prose last.
        out = {
            "metric": metric_name(n_global),
            "value": round(clips_per_s, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
        }
        modes = {}
        if f32_mode:
            modes["f32_mfma"] = {"clips_s": f32_mode["value"], "ms": f32_mode["ms_per_step"]}
        if f16x2_mode:
            modes["f16x2"] = {"clips_s": f16x2_mode["value"], "ms": f16x2_mode["ms_per_step"]}
        if bf16_mode:
            modes["bf16_config5"] = {"clips_s": bf16_mode["value"], "ms": bf16_mode["ms_per_step"]}
        if modes:
            out["modes"] = modes
        if inference:
            out["inference"] = inference
        if coll:
            out["collective"] = coll
        out["step_fractions"] = {
            "mfma_f32": round(flops / (elapsed / args.steps) / world / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
            "mfma_of_this_math_mode": round(flops / (elapsed / args.steps) / world / (MATH_PEAK[args.math] * 1e12), 4),
            "hbm": round(byts / (elapsed / args.steps) / world / (PEAK_HBM_GBPS * 1e9), 4),
            # ... against the rate a streaming kernel reaches on this part (MI355X_MICROARCH.md: ~6.3 TB/s of the 8.0 TB/s spec)
            "hbm_of_achievable_6300": round(byts / (elapsed / args.steps) / world / (ACHIEVABLE_HBM_GBPS * 1e9), 4),
            "algorithmic_gflop_per_clip": round(flops / n_global / 1e9, 2),
            "algorithmic_mb_per_clip": round(byts / n_global / 1e6, 2)}
        if args.math == "bf16":      # the activation-sized tensors of this mode are two bytes wide: SURVEY 8d's byte count halves
            out["step_fractions"]["hbm_bf16_storage"] = round(0.5 * byts / (elapsed / args.steps) / world / (PEAK_HBM_GBPS * 1e9), 4)
            out["step_fractions"]["algorithmic_mb_per_clip_bf16_storage"] = round(0.5 * byts / n_global / 1e6, 2)
        notes = {}
        roofline_detail = None
        if kern:
            dom = max(kern, key=lambda k: k["channels"])      # the 256-channel launch: the longest one of the step
            peak = MATH_PEAK[args.math]
            kname = MATH_KERNEL[args.math].format(nt=2 if dom["channels"] <= 64 else 4, nt2=1 if dom["channels"] <= 64 else 2)
            out["roofline"] = {"bound": "mfma", "achieved": round(dom["tflops"], 2), "peak": round(peak, 1),
                               "unit": "TFLOP/s", "frac": round(dom["tflops"] / peak, 4),
                               "traffic": measured_traffic(dom, n_local * SHAPE["M"], args.math),
                               "ms_per_launch": round(dom["ms"], 4),
                               "flop_per_launch": dom["flops"],
                               "frac_of_f32_mfma_peak": round(dom["tflops"] / PEAK_F32_MFMA_TFLOPS, 4),
                               "kernel": f"{kname} (9x1 temporal conv forward, "
                                         f"{dom['channels']} channels, {dom['frames']} frames)"}
            notes["roofline.traffic"] = ("stored rocprofv3 PMC record of this kernel at this shape (profiles/r0*_traffic.json: "
                                         "2*FETCH_SIZE + WRITE_SIZE per launch), not measured in this run")
            if world == 1 and not args.no_live_traffic and not args.no_kernel_timing:
                log("measuring the dominant kernel's HBM-side traffic (two rocprofv3 --pmc child passes)")
                live, how = live_traffic(args.math, args.batch)
                if live is not None:
                    out["roofline"]["traffic_stored_record"] = out["roofline"]["traffic"]
                    out["roofline"]["traffic"] = live
                    notes["roofline.traffic"] = how
                else:
                    notes["roofline.traffic"] += f" (live measurement failed: {how})"
            notes["roofline.practical_ceiling"] = ("MI355X_MICROARCH.md: tuned bf16 MFMA loops reach 1.25-1.48 PFLOP/s on random data "
                                                   "(the chip lowers its clock to ~1.9 GHz under MFMA load); this kernel issues "
                                                   "6 x achieved of bf16 MFMA work")
            notes["roofline.peak"] = {"f32": "v_mfma_f32_32x32x2_f32 dense", "bf16": "v_mfma_f32_32x32x16_bf16 dense",
                                      "bf16x3": "bf16 dense peak / 6 partial products (f32-equivalent FLOPs)",
                                      "f16x2": "f16 dense peak (= bf16's) / 3 products (f32-equivalent FLOPs)"}[args.math]
            roofline_detail = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()} for d in kern]
        parity = None
        if not args.no_cpu_baseline:
            # rank 0 only, after every timed region (N > 1: the other ranks wait at the barrier below; nothing here is a collective)
            log("timing the CPU oracle on the host cores")
            also = ("f16x2",) if f16x2_mode else ()
            out["cpu_baseline"], parity = cpu_baseline(device=device, maths=(args.math,) + also)
        out["dtype"] = MATH_DTYPE_SHORT[args.math]
        out["data"] = "synthetic"
        out["config"] = {"workload": "AGCN 10-block fwd+bwd, %s, synthetic (N,C,T,V,M)=(%d,3,300,%d,2), "
                                     "%d classes, train-mode BatchNorm, CrossEntropy, all parameter gradients"
                                     % ({25: "NTU-RGB-D graph", 27: "NTU-RGB-D graph + 2 IMU joints", 22: "MMAct COCO-18 graph + 4 IMU joints"}
                                        [SHAPE["V"]], n_global, SHAPE["V"], SHAPE["classes"]),
                         "global_batch": n_global, "per_gpu_batch": n_local,
                         "parallelism": f"dp{world}", "launch": mode, "loss": round(loss_val, 5),
                         "optimizer_step_in_timed_region": args.optimizer, "input_pipeline": pipeline or "one HBM-resident batch",
                         "init_override": "gcn1.bn.weight=1.0 (reference: 1e-6); see notes"}
        if other:
            out["other_scaling"] = other
        if parity is not None:
            out["parity_at_full_shape"] = parity[args.math]
            if f16x2_mode:
                f16x2_mode["parity_at_full_shape"] = parity["f16x2"]
        if f32_mode:
            out["f32_mfma_mode"] = f32_mode
        if f16x2_mode:
            out["f16x2_mode"] = f16x2_mode
        if bf16_mode:
            out["bf16_mode"] = bf16_mode
        if roofline_detail:
            out["roofline_all_widths"] = roofline_detail
        notes["dtype"] = MATH_DTYPE[args.math]
        notes["init_override"] = ("gcn1.bn.weight=1.0 in all ten blocks (the reference initialises it to 1e-6, agcn.py:86-94; O(1) "
                                  "values give every kernel realistic magnitudes; everything else torch.manual_seed(1) reference init)")
        out["notes"] = notes
        print(json.dumps(out), flush=True)
    if world > 1:
        wait_for_rank0(rank)      # rank 0 has timed the CPU oracle meanwhile: the others poll the rendezvous store, not a collective
        dist.barrier()
        dist.destroy_process_group()


def wait_for_rank0(rank: int, key: str = "fgcn_bench_rank0_done", timeout_s: int = 7200) -> None:
    """After the timed regions rank 0 alone times the CPU oracle (tens of seconds to minutes on a slow host) and prints the line.  The other
    ranks must not sit in a collective meanwhile (an NCCL barrier is a GPU operation under the process group's watchdog timeout): they
    wait on a key of the rendezvous store, which rank 0 sets when it is done."""
    import datetime
    store = dist.distributed_c10d._get_default_store()
    if rank == 0:
        store.set(key, "1")
    else:
        store.wait([key], datetime.timedelta(seconds=timeout_s))


if __name__ == "__main__":
    main()
