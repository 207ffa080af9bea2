#!/usr/bin/env python3
"""Tuning aid: ModifiedResNet (RN50 shape) image tower throughput; run under rocprofv3 for the per-kernel split."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import synthetic as syn
from clip_calibration_amd.model import build_model
sd = syn.synthetic_resnet_state_dict((3, 4, 6, 3), 64, 224, "RN50", seed=0)
model = build_model(sd, None).cuda()
B = int(os.environ.get("B", "256"))
img = syn.synthetic_images(B, "RN50", device="cuda")
for _ in range(2):
    model.image_features_f32(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("ITERS", "5"))
for _ in range(n):
    model.image_features_f32(img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"RN50 B={B}: {dt*1e3:.2f} ms  {B/dt:.0f} img/s", flush=True)
