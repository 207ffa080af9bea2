#!/usr/bin/env python3
"""Tuning aid: interleaved A/B of several (option = value) CONFIGURATIONS on the per-layer kernels and the whole image tower.
    CONFIGS="gemm_stream=0,gemm_rstream=0;gemm_stream=1,gemm_rstream=1" python tools/block_ab2.py"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model
CONFIGS = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",")) for c in os.environ["CONFIGS"].split(";")]
B = int(os.environ.get("B", "256")); ROUNDS = int(os.environ.get("ROUNDS", "5")); G = os.environ.get("GEOM", "ViT-B/16")
model = build_model(dict(syn.synthetic_state_dict(G, seed=0)), None).cuda()
images = syn.synthetic_images(B, G, seed=0, device="cuda")
per = [{k: [] for k in model.BLOCK_KERNELS} for _ in CONFIGS]; tower = [[] for _ in CONFIGS]; ref = None
with torch.no_grad():
    for r in range(ROUNDS + 1):
        for ci, cfg in enumerate(CONFIGS):
            for k, v in cfg.items():
                _lib.set_option(k, v)
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
            if r == 0:
                fn, rn = torch.nn.functional.normalize(f, dim=1), torch.nn.functional.normalize(ref, dim=1)
                print(f"{cfg}: max |d cos| vs first config {float((fn - rn).abs().max()):.2e}", flush=True)
            else:
                tower[ci].append(e0.elapsed_time(e1) / 3)
                for k in ms:
                    per[ci][k].append(ms[k] * 1e3)
for ci, cfg in enumerate(CONFIGS):
    print(f"{cfg}: tower med {statistics.median(tower[ci]):.3f} ms (min {min(tower[ci]):.3f}) | " +
          " | ".join(f"{k} {statistics.median(t):6.1f}" for k, t in per[ci].items()), flush=True)
