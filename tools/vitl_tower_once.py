#!/usr/bin/env python3
"""Ten passes of the ViT-L/14@336px image tower at its per-rank batch (BASELINE configs[4]: 64 images per GPU; MODEL / BATCH override): the program
rocprofv3 --kernel-trace --stats wraps for profiles/r05_vitl336_kernel_stats.csv."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import synthetic as syn
from clip_calibration_amd.model import build_model
gname, B = os.environ.get("MODEL", "ViT-L/14@336px"), int(os.environ.get("BATCH", "64"))
model = build_model(syn.synthetic_state_dict(gname), None).cuda()
img = syn.synthetic_images(B, gname, device="cuda")
with torch.no_grad():
    for _ in range(int(os.environ.get("PASSES", "10"))):
        out = model.image_features_f32(img)
torch.cuda.synchronize()
assert torch.isfinite(out).all()
