#!/usr/bin/env python3
"""Tuning aid: image-tower throughput vs batch size (does a batch whose activations fit the 256 MiB Infinity Cache run
faster per image than batch 256 streaming through HBM?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model

sd = syn.synthetic_state_dict("ViT-B/16")
model = build_model(dict(sd), None).cuda()
for v in os.environ.get("VARIANTS", "auto,0,1").split(","):
    if v == "auto":
        _lib.set_option("gemm_variant", -1)
    else:
        _lib.set_option("gemm_variant", _lib.gemm_variant_id(v))
    row = [f"variant {v:4s}:"]
    for B in (16, 32, 48, 64, 96, 128, 256):
        img = syn.synthetic_images(B, "ViT-B/16", device="cuda")
        for _ in range(3):
            model.image_features_f32(img)
        torch.cuda.synchronize()
        n = max(4, 512 // B)
        t0 = time.perf_counter()
        for _ in range(n):
            model.image_features_f32(img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        row.append(f" B={B}: {B/dt:7.0f} img/s")
    print("".join(row), flush=True)
