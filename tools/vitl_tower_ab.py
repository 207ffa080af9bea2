#!/usr/bin/env python3
"""Interleaved A/B rounds (one process) of the ViT-L image towers at their per-rank shapes (BASELINE configs[4]: ViT-L/14@336px, 64 images per GPU;
ViT-L/14 at 128) with the long-sequence attention on the round-1 streaming kernel (attn_ring 0) and on the ring kernel (attn_ring 1): the tower
half of the gate of profiles/r05_vitl_attention.txt (the kernel half: tools/attn_ring_ab.py).  Synthetic weights, events on the launch stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model

def dev_ms(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for gname, B in (("ViT-L/14@336px", 64), ("ViT-L/14", 128)):
    model = build_model(syn.synthetic_state_dict(gname), None).cuda()
    img = syn.synthetic_images(B, gname, device="cuda")
    outs, res = {}, {0: [], 1: []}
    with torch.no_grad():
        for m in (0, 1):
            _lib.set_option("attn_ring", m)
            outs[m] = model.image_features_f32(img).clone()
        for rnd in range(5):
            for m in (0, 1):
                _lib.set_option("attn_ring", m)
                res[m].append(dev_ms(lambda: model.image_features_f32(img), 4))
    a, b = torch.nn.functional.normalize(outs[0], dim=1), torch.nn.functional.normalize(outs[1], dim=1)
    cosd = float((1.0 - (a * b).sum(1)).abs().max())
    med = {m: sorted(v)[2] for m, v in res.items()}
    fl = syn.flops_per_image(gname) * B
    print(f"{gname} B={B}: " + " | ".join(f"attn_ring {m}: {med[m]:7.3f} ms = {B / med[m] * 1e3:7.0f} img/s ({fl / med[m] / 1e9:5.0f} TF)" for m in (0, 1)) +
          f" | tower {100 * (med[0] / med[1] - 1):+.1f} % | max |1 - cos| between the two towers' features {cosd:.2e}", flush=True)
    del model, img
    torch.cuda.empty_cache()
_lib.set_option("attn_ring", 1)
