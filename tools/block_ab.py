#!/usr/bin/env python3
"""Tuning aid: interleaved A/B (one process, cdna_hip_programming.md rule 24) of a runtime option on the five per-layer
kernels of the image tower as the tower launches them (clipmi_profile_block) and on the whole tower.

    OPTION=gemm_stream VALUES=0,1 python tools/block_ab.py"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model

OPTION = os.environ.get("OPTION", "gemm_stream")
VALUES = [int(v) for v in os.environ.get("VALUES", "0,1").split(",")]
B = int(os.environ.get("B", "256"))
ROUNDS = int(os.environ.get("ROUNDS", "6"))
G = os.environ.get("GEOM", "ViT-B/16")
model = build_model(dict(syn.synthetic_state_dict(G, seed=0)), None).cuda()
images = syn.synthetic_images(B, G, seed=0, device="cuda")
per = {v: {k: [] for k in model.BLOCK_KERNELS} for v in VALUES}
tower = {v: [] for v in VALUES}
ref = None
with torch.no_grad():
    for r in range(ROUNDS + 1):
        for v in VALUES:
            _lib.set_option(OPTION, v)
            f = model.image_features_f32(images)
            if ref is None:
                ref = f.clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f = model.image_features_f32(images)
            e1.record(); torch.cuda.synchronize()
            ms = model.profile_block_ms(B, iters=8)
            if r > 0:
                tower[v].append(e0.elapsed_time(e1) / 3)
                for k in ms:
                    per[v][k].append(ms[k] * 1e3)
            if r == 0:
                fn, rn = torch.nn.functional.normalize(f, dim=1), torch.nn.functional.normalize(ref, dim=1)
                print(f"{OPTION}={v}: max |d cos| of the image features vs {OPTION}={VALUES[0]}: {float((fn - rn).abs().max()):.2e}", flush=True)
for v in VALUES:
    print(f"{OPTION}={v}: tower med {statistics.median(tower[v]):.3f} ms (min {min(tower[v]):.3f}) | " +
          " | ".join(f"{k} {statistics.median(t):6.1f} us" for k, t in per[v].items()), flush=True)
