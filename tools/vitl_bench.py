#!/usr/bin/env python3
"""Tuning aid: ViT-L/14@336 image tower (BASELINE configs[4] per-GPU batch 64) under forced GEMM variants."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model
G = os.environ.get("GEOM", "ViT-L/14@336px")
B = int(os.environ.get("B", "64"))
model = build_model(syn.synthetic_state_dict(G), None).cuda()
img = syn.synthetic_images(B, G, device="cuda")
for v in os.environ.get("VARIANTS", "auto,1,a,auto,a").split(","):
    if v == "auto":
        _lib.set_option("gemm_variant", -1)
    else:
        _lib.set_option("gemm_variant", _lib.gemm_variant_id(v))
    for _ in range(2):
        model.image_features_f32(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        model.image_features_f32(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print(f"{G} B={B} variant {v:4s}: {B/dt:8.0f} img/s  {syn.flops_per_image(G)*B/dt/1e12:6.0f} TFLOP/s", flush=True)
