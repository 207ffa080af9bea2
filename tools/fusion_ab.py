#!/usr/bin/env python3
"""Interleaved A/B of the fused in-projection + attention kernel (option attn_loader = 3, clipmi_qkv_attention) against the two launches it
replaces (in-projection GEMM + attention_vision_nt_kernel, attn_loader = 2), on the image tower's own operands at batch 256 (ViT-B/16):
  pair, back to back   clipmi_profile_block steps 0 + 1 (each kernel in a loop of its own launches)
  pair, in the tower   clipmi_encode_image_timed: the two intervals of every layer of real passes
  tower                the whole image tower, torch events
Gate (task 2): adopt only at >= 8 % on the pair.       python tools/fusion_ab.py [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clip_calibration_amd import _lib, synthetic as syn  # noqa: E402
from clip_calibration_amd.model import build_model  # noqa: E402


def timed_ms(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = 256
    sd = syn.synthetic_state_dict("ViT-B/16", seed=0)
    model = build_model(dict(sd), {"trainer": "ZeroshotCLIP"}).cuda()
    images = syn.synthetic_images(B, "ViT-B/16", seed=0, device=torch.device("cuda"))
    rows = {2: {"pair_b2b": [], "pair_tower": [], "tower": []}, 3: {"pair_b2b": [], "pair_tower": [], "tower": []}}
    with torch.no_grad():
        for r in range(rounds + 1):
            for mode in (2, 3):
                _lib.set_option("attn_loader", mode)
                model.image_features_f32(images)
                ms = model.profile_block_ms(B, iters=12)
                t = [model.image_tower_launch_us(images) for _ in range(2)][-1]
                tower = timed_ms(lambda: model.image_features_f32(images), 6)
                if r == 0:
                    continue     # warm-up round
                rows[mode]["pair_b2b"].append(1e3 * (ms["in_proj"] + ms["attention"]))
                rows[mode]["pair_tower"].append(float(np.mean([b[0] + b[1] for b in t["blocks"]])))
                rows[mode]["tower"].append(tower)
    _lib.set_option("attn_loader", 2)
    print(f"ViT-B/16, batch {B}, {rounds} interleaved rounds (median [min .. max])")
    for key, unit in (("pair_b2b", "us"), ("pair_tower", "us"), ("tower", "ms")):
        a, b = np.array(rows[2][key]), np.array(rows[3][key])
        print(f"  {key:11s} two launches {np.median(a):8.2f} [{a.min():.2f} .. {a.max():.2f}] {unit} | fused {np.median(b):8.2f} [{b.min():.2f} .. {b.max():.2f}] {unit}"
              f" | fused / two = {np.median(b) / np.median(a):.3f}")


if __name__ == "__main__":
    main()
