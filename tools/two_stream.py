#!/usr/bin/env python3
"""Experiment: two half-batches on two HIP streams (does the HBM-bound work of one hide under the MFMA-bound work of
the other?) vs one stream with the full batch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import synthetic as syn
from clip_calibration_amd.model import build_model

sd = syn.synthetic_state_dict("ViT-B/16")
models = [build_model(dict(sd), None).cuda() for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for B in (128, 256):
    imgs = [syn.synthetic_images(B, "ViT-B/16", seed=i, device="cuda") for i in range(2)]
    def one_stream(n):
        for _ in range(n):
            models[0].image_features_f32(imgs[0]); models[0].image_features_f32(imgs[1])
    def two_streams(n):
        for _ in range(n):
            for m, s, x in zip(models, streams, imgs):
                with torch.cuda.stream(s):
                    m.image_features_f32(x)
    for name, fn in (("1 stream ", one_stream), ("2 streams", two_streams)):
        fn(2); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(6); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 6
        print(f"B={B} x2  {name}: {2*B/dt:8.0f} img/s", flush=True)
