#!/usr/bin/env python3
"""The workload tools/measure_traffic.sh profiles: the IMAGE tower only (ViT-B/16, batch 256), RUNS passes -- no text tower, so every
GEMM dispatch in the counter CSVs has the bench shape (M = 50432) and each role shows up exactly 12 * RUNS times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn  # noqa: E402
from clip_calibration_amd.model import build_model  # noqa: E402

RUNS = int(os.environ.get("RUNS", "3"))
_lib.set_option("cls_only_last_block", int(os.environ.get("CLS_ONLY", "0")))   # every row of every block: all twelve layers launch the same five shapes
model = build_model(dict(syn.synthetic_state_dict("ViT-B/16", seed=0)), {"trainer": "ZeroshotCLIP"}).cuda()
images = syn.synthetic_images(256, "ViT-B/16", seed=0, device="cuda")
with torch.no_grad():
    for _ in range(RUNS):
        model.image_features_f32(images)
torch.cuda.synchronize()
print("runs", RUNS)
