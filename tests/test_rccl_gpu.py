"""The RCCL path on the one GPU a test box has (SURVEY.md section 8 row e): a fresh child process initialises the `nccl`
backend with world size 1, replays the recorded training step of the real model and all-reduces its 13.9 MB flat gradient
buffer after every replay (tools/rccl_world1_check.py).  Two ranks on one device are never started with `nccl`; the N > 1
control flow is covered with gloo (tests/test_dp_gloo.py, tests/test_bench_host.py, tests/test_session_gpu.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_world1_rccl_all_reduce_behind_graph_replay():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1_check.py"), "--steps", "12"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps(rec))
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["flat_bytes"] >= 13_800_000                      # the real exchange buffer (3.47 M float32 parameters)
    assert rec["replays"] == 12 and rec["buffer_unchanged_by_the_collective"] and rec["loss_equal"]
    assert rec["flat_grad_rel_l2_vs_eager_without_group"] < 1e-5
