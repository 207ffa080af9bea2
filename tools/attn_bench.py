#!/usr/bin/env python3
"""Tuning aid: attention kernel time at the tower shapes (persistent vs per-item kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
from gemm_bench import timeit

for name, n, l, h, causal in [("vision B=256", 256, 197, 12, False), ("text C=1000", 1000, 77, 8, True), ("ViT-L@336 B=64", 64, 577, 16, False), ("ViT-L B=64", 64, 257, 16, False)]:
    qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
    row = [f"{name:14s}"]
    for np_ in ("0", "1"):
        _lib.set_option("attn_no_persist", int(np_))
        t = timeit(lambda: ops.attention(qkv, n, l, h, causal))
        gb = (qkv.numel() + n * l * 64 * h) * 2 / 1e9
        row.append(f" {'persist' if np_ == '0' else 'per-item'}: {t*1e3:7.1f} us ({gb/t*1e3/1e3:5.2f} TB/s)")
    print("".join(row), flush=True)
