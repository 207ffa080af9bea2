#!/usr/bin/env python3
"""Diagnostic (never a benchmark): per-tile stamps of gemm_rstream_kernel as the image tower launches it (out-proj = step 2,
c_proj = step 4 of clipmi_profile_block).  Needs the tuning build:

    make -C clip_calibration_amd/csrc tuning
    CLIPMI_LIBRARY=clip_calibration_amd/csrc/libclipmi_tuning.so python tools/rstream_stamps.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn  # noqa: E402
from clip_calibration_amd.model import build_model  # noqa: E402

if not hasattr(_lib.lib, "clipmi_tuning_set_stamps"):
    raise SystemExit("libclipmi.so is the product build: `make -C clip_calibration_amd/csrc tuning` and set CLIPMI_LIBRARY")
_lib.lib.clipmi_tuning_set_stamps.argtypes = [ctypes.c_void_p]

B = int(os.environ.get("B", "256"))
model = build_model(dict(syn.synthetic_state_dict("ViT-B/16", seed=0)), None).cuda()
images = syn.synthetic_images(B, "ViT-B/16", seed=0, device="cuda")
with torch.no_grad():
    for _ in range(20):
        model.image_features_f32(images)
torch.cuda.synchronize()
for name, step in (("out_proj", 2), ("c_proj", 4)):
    stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
    model.profile_block_ms(B, iters=3, only=step)
    _lib.lib.clipmi_tuning_set_stamps(stamps.data_ptr())
    ms = model.profile_block_ms(B, iters=1, only=step)[model.BLOCK_KERNELS[step]]
    torch.cuda.synchronize()
    _lib.lib.clipmi_tuning_set_stamps(None)
    raw = stamps.cpu().numpy().reshape(-1, 8)
    s = raw[:1024]
    live = s[:, 0] > 0
    t0 = s[live, 0].min()
    t = (s[:, :5] - t0) / 100.0
    print(f"{name}: hipEvents {ms * 1e3:.1f} us; {int(live.sum())} tiles stamped; first tile start spread {np.ptp(t[live & (np.arange(1024) % 4 == 0), 0]):.2f} us; "
          f"last tile end (closing barrier) {t[live, 4].max():.1f} us")
    for k in range(4):
        rows = np.arange(k, 1024, 4)
        rows = rows[live[rows]]
        if not len(rows):
            continue
        tk, sk = t[rows], s[rows]
        clk = (sk[:, 7] - sk[:, 6]) / np.maximum(sk[:, 2] - sk[:, 1], 1) * 100.0
        print(f"  tile {k}: {len(rows)} tiles | start med {np.median(tk[:, 0]):6.1f} us | K-step 0 {np.median(tk[:, 1] - tk[:, 0]):5.2f} | rest of the K loop {np.median(tk[:, 2] - tk[:, 1]):6.2f} "
              f"| convert + store + next residual {np.median(tk[:, 3] - tk[:, 2]):5.2f} (p90 {np.percentile(tk[:, 3] - tk[:, 2], 90):5.2f}) "
              f"| wait + barrier {np.median(tk[:, 4] - tk[:, 3]):5.2f} (p90 {np.percentile(tk[:, 4] - tk[:, 3], 90):5.2f}) | clock {np.median(clk):.0f} MHz")
