#!/usr/bin/env python3
"""Record throughput of the other BASELINE configs' per-GPU shapes (image tower only, synthetic weights)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import synthetic as syn
from clip_calibration_amd.model import build_model

for gname, batches in (("ViT-B/16", (128, 1024, 2048)), ("ViT-L/14", (64, 256)), ("ViT-L/14@336px", (64, 128))):
    sd = syn.synthetic_state_dict(gname)
    model = build_model(sd, None).cuda()
    del sd
    for B in batches:
        img = syn.synthetic_images(B, gname, device="cuda")
        out = model.image_features_f32(img); torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        n = 3
        t0 = time.perf_counter()
        for _ in range(n):
            model.image_features_f32(img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{gname:16s} B={B:5d}: {B/dt:8.0f} img/s  {syn.flops_per_image(gname)*B/dt/1e12:6.0f} TFLOP/s", flush=True)
    del model
    torch.cuda.empty_cache()

# ModifiedResNet (RN50) image tower: im2col + GEMM convolutions, not tuned -- a correctness-first tower (SURVEY f-4)
sd = syn.synthetic_resnet_state_dict((3, 4, 6, 3), 64, 224, "RN50", seed=0)
model = build_model(sd, None).cuda()
for B in (64, 256):
    img = syn.synthetic_images(B, "RN50", device="cuda")
    model.image_features_f32(img); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        model.image_features_f32(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"RN50             B={B:5d}: {B/dt:8.0f} img/s", flush=True)
