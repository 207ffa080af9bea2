#!/usr/bin/env python3
"""Where do the joules of a GEMM launch go?  One process per library variant (CLIPMI_LIBRARY = the tuning build or one of the
`make ablate` builds; KNOB = runtime ablation bits of the tuning build): the image tower's c_fc (step 3) and c_proj (step 4) launched
back to back on the tower's own operands for SECONDS each while a thread samples package power and sclk from sysfs (bench.PowerSampler).
Prints  variant | kernel | us per launch | W | sclk MHz | mJ per launch.   Driver: tools/energy_attribution.sh"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PowerSampler  # noqa: E402
from clip_calibration_amd import _lib, synthetic as syn  # noqa: E402
from clip_calibration_amd.model import build_model  # noqa: E402

B = int(os.environ.get("B", "256"))
SECONDS = float(os.environ.get("SECONDS", "2.5"))
knob = int(os.environ.get("KNOB", "0"))
name = os.environ.get("VARIANT", os.path.basename(os.environ.get("CLIPMI_LIBRARY", "libclipmi.so")))
if knob:
    _lib.lib.clipmi_tuning_set_knob.argtypes = [ctypes.c_int]
model = build_model(dict(syn.synthetic_state_dict("ViT-B/16", seed=0)), None).cuda()
images = syn.synthetic_images(B, "ViT-B/16", seed=0, device="cuda")
with torch.no_grad():
    for _ in range(3):
        model.image_features_f32(images)          # real activations in the workspace
torch.cuda.synchronize()
for kernel, step in (("c_fc", 3), ("c_proj", 4), ("in_proj", 0), ("out_proj", 2)):
    if os.environ.get("KERNELS") and kernel not in os.environ["KERNELS"].split(","):
        continue
    if knob:
        _lib.lib.clipmi_tuning_set_knob(knob)
    model.profile_block_ms(B, iters=20, only=step)
    sampler = PowerSampler(0, period_s=0.02)
    sampler.start()
    t_end, ms, n = time.time() + SECONDS, [], 0
    while time.time() < t_end:
        ms.append(model.profile_block_ms(B, iters=200, only=step)[kernel])
        n += 1
    pw = sampler.stop()
    if knob:
        _lib.lib.clipmi_tuning_set_knob(0)
    us = 1e3 * sum(ms) / len(ms)
    w, f = pw.get("avg_w"), pw.get("sclk_mhz_avg")
    print(f"{name:34s} | {kernel:8s} | {us:7.1f} us | {w if w is None else round(w):>5} W | {f if f is None else round(f):>5} MHz | "
          f"{'' if w is None else f'{w * us * 1e-3:7.1f}'} mJ per launch", flush=True)
    with torch.no_grad():
        model.image_features_f32(images)          # ablated launches leave garbage in the workspace: restore real activations
