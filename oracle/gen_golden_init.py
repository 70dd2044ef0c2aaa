#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generate tests/golden/init_and_format.json by IMPORTING THE REFERENCE (build container only).

    python oracle/gen_golden_init.py

Two pins that need the reference itself rather than the oracle (import recipe: oracle/gen_golden.py):

  * initialisation (reference torch_src/models/mmargcn/agcn.py:18-34,62-63,86-94,179 and
    torch_src/models/agcn/agcn.py): the reference's models are constructed under ``torch.manual_seed(1)`` (every config's
    ``fixed_seed: 1``) and each state-dict entry is fingerprinted (numel, sum, sum of squares, first / last element, float64)
    in state-dict ORDER.  A model built here from the same seed must reproduce the fingerprints -- same draws from the same
    generator in the same order -- and the key order.
  * feature-file format (row f2; util/preprocessing/data_writer.py:11-38,71-83): a tiny feature file is written with the
    reference's own ``NumpyWriter`` and its bytes are stored (hex) with their SHA-256; fusion_gcn_amd.data.NumpyWriter
    must produce the same bytes.
Only data is stored: numbers, key names, and the bytes of a generated file.
"""
import hashlib
import json
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np  # noqa: E402

np.int = int
np.float = float
_tv = types.ModuleType("torchvision")
_tv.models = types.ModuleType("torchvision.models")
sys.modules.update({"torchvision": _tv, "torchvision.models": _tv.models, "cv2": types.ModuleType("cv2")})

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FGCN_REFERENCE", "/root/reference")
sys.path[:0] = [REF, os.path.join(REF, "torch_src")]
sys.path.append(REPO)

import torch  # noqa: E402

from util.graph import Graph  # noqa: E402  (reference)
import models.mmargcn.agcn as ref_agcn  # noqa: E402  (reference)
import models.mmargcn.mmargcn as ref_mm  # noqa: E402  (reference)
import models.agcn.agcn as ref_agcn_cuda  # noqa: E402  (reference)
from datasets.ntu_rgb_d import constants as ntu  # noqa: E402  (reference)
from datasets.utd_mhad import constants as utd  # noqa: E402  (reference)
from util.preprocessing.data_writer import NumpyWriter  # noqa: E402  (reference)

from oracle import filler  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "init_and_format.json")
SEED = 1


def fingerprint(state_dict):
    rows = []
    for k, v in state_dict.items():
        t = v.detach().double().flatten()
        rows.append([k, list(v.shape), float(t.sum()), float((t * t).sum()),
                     float(t[0]) if t.numel() else 0.0, float(t[-1]) if t.numel() else 0.0])
    return rows


def main():
    out = {"torch_version": torch.__version__, "seed": SEED, "init": {}}
    g_ntu = Graph(ntu.skeleton_edges, center_joint=ntu.center_joint)
    g_utd = Graph(utd.skeleton_edges, center_joint=utd.center_joint)
    cases = {
        "mmargcn.agcn/ntu60": lambda: ref_agcn.Model((2, 300, 25, 3), 60, g_ntu),
        "mmargcn.agcn/utd27_dropout": lambda: ref_agcn.Model((1, 100, 20, 3), 27, g_utd, dropout=0.25),
        "mmargcn.agcn/utd_6layers_nofc": lambda: ref_agcn.Model((1, 100, 20, 3), 27, g_utd, num_layers=6, without_fc=True),
        "agcn/utd27": lambda: ref_agcn_cuda.Model({"skeleton": (1, 100, 20, 3)}, 27, g_utd),
        "mmargcn/skeleton_imu_spatial_fusion": lambda: ref_mm.Model(
            {"skeleton": (1, 20, 22, 3)}, 27, g_utd, mode="skeleton_imu_spatial_fusion", num_imu_joints=2,
            imu_enhanced_mode="append_center"),
    }
    for tag, build in cases.items():
        torch.manual_seed(SEED)
        out["init"][tag] = fingerprint(build().state_dict())

    # feature file written by the reference's NumpyWriter
    shape = (3, 2, 4, 5, 3)
    feats = filler.skeleton_input("x.featfile", shape).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "skeleton_train_features.npy")
        with NumpyWriter(path, np.float32, shape) as w:
            for s in feats:
                w.collect_next(s)
        raw = open(path, "rb").read()
    out["feature_file"] = {"writer": "util/preprocessing/data_writer.py NumpyWriter", "input": "filler.skeleton_input('x.featfile', shape) as float32",
                           "shape": list(shape), "dtype": "float32", "size": len(raw), "sha256": hashlib.sha256(raw).hexdigest(),
                           "bytes_hex": raw.hex()}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0)
    print(OUT, os.path.getsize(OUT), "bytes;", {k: len(v) for k, v in out["init"].items()})


if __name__ == "__main__":
    main()
