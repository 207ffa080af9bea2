#!/usr/bin/env python3
"""Measurement: error of the long-sequence attention kernels (ring: attn_ring 1, streaming: attn_ring 0) against an fp32 reference of the same fp16 inputs, at
several score magnitudes (scale of the random q / k / v).  CLIPMI_LIBRARY selects the build (profiles/r05_ring_waves.txt: the pre-scaled-Q form lost here)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
def ref_attn(qkv, n, l, h):
    q, k, v = qkv.float().view(n, l, 3, h, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(n * l, h * 64)
for (n, l, h, scale) in [(2, 257, 16, 1.5), (1, 577, 4, 1.5), (2, 480, 2, 1.5), (1, 1025, 1, 1.5), (2, 577, 16, 3.0), (2, 577, 16, 0.5)]:
    g = torch.Generator().manual_seed(n * 1000 + l + h)
    qkv = (torch.randn(n * l, 3 * 64 * h, generator=g) * scale).half()
    got = ops.attention(qkv.cuda(), n, l, h, False).float().cpu()
    _lib.set_option("attn_ring", 0)
    base = ops.attention(qkv.cuda(), n, l, h, False).float().cpu()
    _lib.set_option("attn_ring", 1)
    ref = ref_attn(qkv, n, l, h)
    print(f"{os.path.basename(os.environ.get('CLIPMI_LIBRARY','libclipmi.so')):22s} n={n} l={l} h={h} scale {scale}: ring max err {float((got-ref).abs().max()):.2e} rms {float((got-ref).pow(2).mean().sqrt()):.2e} | streaming max err {float((base-ref).abs().max()):.2e} rms {float((base-ref).pow(2).mean().sqrt()):.2e} | ring vs streaming {float((got-base).abs().max()):.2e}")
