#!/usr/bin/env python3
"""Launch time of the vision attention kernel inside the ViT-B/16 tower (clipmi_profile_block: hipEvents around back-to-back launches on
the tower's own qkv, no stamps) for the library named by CLIPMI_LIBRARY.  Driver: tools/attn_ablate.sh"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import synthetic as syn  # noqa: E402
from clip_calibration_amd.model import build_model  # noqa: E402

B = int(os.environ.get("B", "256"))
model = build_model(dict(syn.synthetic_state_dict("ViT-B/16", seed=0)), None).cuda()
images = syn.synthetic_images(B, "ViT-B/16", seed=0, device="cuda")
with torch.no_grad():
    for _ in range(3):
        model.image_features_f32(images)
torch.cuda.synchronize()
model.profile_block_ms(B, iters=50, only=1)
us = sorted(1e3 * model.profile_block_ms(B, iters=200, only=1)["attention"] for _ in range(5))
print(f"{os.path.basename(os.environ.get('CLIPMI_LIBRARY', 'libclipmi.so')):24s} {os.environ.get('WHAT', ''):44s} attention {us[2]:6.1f} us (min {us[0]:.1f})", flush=True)
