"""Host-side pieces of bench.py that need no GPU: the algorithmic cost model behind `step_fractions` / `roofline`
(116.44 GFLOP and 411.42 MB per clip at the headline shape) and the committed PMC traffic record."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_costs_match_the_survey():
    flops, byts = bench.algorithmic_costs(64)
    assert abs(flops / 64 / 1e9 - 116.44) < 0.01
    assert abs(byts / 64 / 1e6 - 411.42) < 0.01
    f1, b1 = bench.algorithmic_costs(1)
    assert abs(f1 * 64 - flops) < 1e-6 * flops and abs(b1 * 64 - byts) < 1e-6 * byts      # linear in the clip count
    # the f32 MFMA roof in clips/s that DESIGN.md quotes
    assert abs(bench.PEAK_F32_MFMA_TFLOPS * 1e12 / (flops / 64) - 1351) < 5


def test_traffic_record_is_for_the_dominant_kernel_shape():
    rec = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))["conv_halo_fwd"]
    dom = dict(channels=256, frames=75)
    assert bench.measured_traffic(dom, 128) == rec["traffic_bytes"] == (2 * rec["fetch_size_kib_raw"] + rec["write_size_kib"]) * 1024
    assert bench.measured_traffic(dom, 16) is None                      # other batch: not the measured shape
    assert bench.measured_traffic(dict(channels=64, frames=300), 128) is None
    assert rec["algorithmic_bytes"] == 4 * 128 * 75 * 25 * 2 * 256


def test_bench_starts_its_own_ranks_and_reports_strong_scaling():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses): bench.py starts the two ranks itself
    (a child torch.distributed.run, 127.0.0.1 rendezvous), shards ONE 64-clip batch over them (strong scaling is the reported
    data-parallel figure) and rank 0 prints one JSON line.  --dry-run: the same control flow over gloo, no GPU."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["FGCN_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["dry_run"] is True
    assert out["config"]["global_batch"] == 64 and out["config"]["per_gpu_batch"] == 32 and out["config"]["parallelism"] == "dp2"
    # the N > 1 line proves which collective ran where: backend, world, every rank's device, the all-reduce timed alone, and it carries
    # the CPU baseline too (rank 0; the dry run does not time it)
    coll = out["collective"]
    assert coll["backend"] == "gloo" and coll["world"] == 2 and [d["rank"] for d in coll["devices"]] == [0, 1]
    assert all(set(d) >= {"rank", "local_device", "name"} for d in coll["devices"])
    assert coll["allreduce_bytes"] == 4 * 3_469_510 and coll["allreduce_ms"] > 0 and coll["allreduce_reps"] >= 1 and "rccl_version" in coll
    assert out["cpu_baseline"]["kind"] == "port"
    assert out["metric"].startswith("clips/sec (N,C,T,V,M)=(64,3,300,25,2)")
    # the clip count in `metric` follows the batch that ran
    r8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "2", "--warmup", "1", "--dry-run"],
                        env=env, capture_output=True, text=True, timeout=300)
    assert r8.returncode == 0, r8.stderr[-2000:]
    out8 = json.loads([ln for ln in r8.stdout.splitlines() if ln.startswith("{")][0])
    assert out8["metric"].startswith("clips/sec (N,C,T,V,M)=(8,3,300,25,2)") and out8["config"]["per_gpu_batch"] == 4
    # a batch that does not divide over the ranks is refused before any work
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "7", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_round2_traffic_record_shows_the_fetch_reduction():
    """profiles/r02_traffic.json holds the dominant kernel's PMC record before and after the XCD-aware workgroup order; bench.py
    reports the newer one as `roofline.traffic` for the split-bf16 arithmetic."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))
    new, old = rec["conv_halo_x3_fwd"], rec["conv_halo_x3_fwd_plain_grid"]
    for r in (new, old):
        assert r["traffic_bytes"] == int((2 * r["fetch_size_kib_raw"] + r["write_size_kib"]) * 1024)
        assert r["algorithmic_bytes"] == 4 * 128 * 75 * 25 * 2 * 256
    assert new["traffic_bytes"] < 0.55 * old["traffic_bytes"] and new["traffic_bytes"] < 2 * new["algorithmic_bytes"]
    # round 3 re-collected the record at its final state, for both product forms of the kernel; bench.py reports the newest
    r3 = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))
    for key in ("conv_halo_x3_fwd", "conv_halo_f16x2_fwd"):
        r = r3[key]
        assert r["traffic_bytes"] == int(2 * r["fetch_size_kib_raw"] * 1024 + r["write_size_kib"] * 1024)
        assert r["algorithmic_bytes"] == 4 * 128 * 75 * 25 * 2 * 256 and r["traffic_bytes"] < 2 * r["algorithmic_bytes"]
    assert r3["conv_halo_f16x2_fwd"]["traffic_bytes"] < r3["conv_halo_x3_fwd"]["traffic_bytes"]      # 4 instead of 6 bytes per weight
    # round 4 re-collected the bf16x3 record (same kernel: within 0.5 % of round 3's); the f16x2 one still comes from round 3's file
    r4 = json.load(open(os.path.join(ROOT, "profiles", "r04_traffic.json")))["conv_halo_x3_fwd"]
    assert r4["traffic_bytes"] == int(round((2 * r4["fetch_size_kib_raw"] + r4["write_size_kib"]) * 1024))
    assert abs(r4["traffic_bytes"] - r3["conv_halo_x3_fwd"]["traffic_bytes"]) < 0.005 * r4["traffic_bytes"]
    # round 5: re-collected once more (within 0.5 % again); it is the stored fallback now -- bench.py measures the figure live in its default run
    r5 = json.load(open(os.path.join(ROOT, "profiles", "r05_traffic.json")))["conv_halo_x3_fwd"]
    assert r5["traffic_bytes"] == int(round((2 * r5["fetch_size_kib_raw"] + r5["write_size_kib"]) * 1024))
    assert abs(r5["traffic_bytes"] - r4["traffic_bytes"]) < 0.005 * r5["traffic_bytes"]
    assert bench.measured_traffic(dict(channels=256, frames=75), 128, "bf16x3") == r5["traffic_bytes"]
    assert bench.measured_traffic(dict(channels=256, frames=75), 128, "f16x2") == r3["conv_halo_f16x2_fwd"]["traffic_bytes"]
