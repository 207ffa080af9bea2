#!/usr/bin/env python3
"""Measurement: the persistent fp16-out GEMM (in-proj / c_fc) at row counts whose ragged last row of 256-row tiles opens a round of its own, with the
remainder rows as a launch of their own (option gemm_split_rows = 1, the default) and in one launch (0); and the ViT-L towers both ways."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops, synthetic as syn
from clip_calibration_amd.model import build_model

def dev_us(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n

g = torch.Generator().manual_seed(0)
for name, M, N, K, epi in (("ViT-L/14@336 x 64 c_fc", 36928, 4096, 1024, _lib.EPI_BIAS_QUICKGELU), ("ViT-L/14@336 x 64 in-proj", 36928, 3072, 1024, _lib.EPI_BIAS),
                           ("ViT-L/14 x 128 c_fc", 32896, 4096, 1024, _lib.EPI_BIAS_QUICKGELU), ("ViT-L/14 x 128 in-proj", 32896, 3072, 1024, _lib.EPI_BIAS),
                           ("ViT-B/16 x 256 c_fc (no remainder)", 50432, 3072, 768, _lib.EPI_BIAS_QUICKGELU)):
    a = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    res = {0: [], 1: []}
    for rnd in range(5):
        for m in (0, 1):
            _lib.set_option("gemm_split_rows", m)
            dev_us(lambda: ops.gemm_f16(a, w, bias, epilogue=epi, out=out), 3)
            res[m].append(dev_us(lambda: ops.gemm_f16(a, w, bias, epilogue=epi, out=out), 20))
    t0, t1 = sorted(res[0])[2], sorted(res[1])[2]
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"{name:36s} M={M} N={N} K={K}: {tiles} tiles = {tiles / 256:.2f} rounds | one launch {t0:7.1f} us | remainder apart {t1:7.1f} us | {100 * (t1 / t0 - 1):+.1f} %", flush=True)

for gname, B in (("ViT-L/14@336px", 64), ("ViT-L/14", 128)):
    model = build_model(syn.synthetic_state_dict(gname), None).cuda()
    img = syn.synthetic_images(B, gname, device="cuda")
    res = {0: [], 1: []}
    with torch.no_grad():
        for rnd in range(5):
            for m in (0, 1):
                _lib.set_option("gemm_split_rows", m)
                model.image_features_f32(img)
                res[m].append(dev_us(lambda: model.image_features_f32(img), 4) / 1e3)
    t0, t1 = sorted(res[0])[2], sorted(res[1])[2]
    print(f"tower {gname} B={B}: one launch {t0:.3f} ms = {B / t0 * 1e3:.0f} img/s | remainder apart {t1:.3f} ms = {B / t1 * 1e3:.0f} img/s | {100 * (t0 / t1 - 1):+.1f} %", flush=True)
_lib.set_option("gemm_split_rows", 1)
