#!/usr/bin/env python3
"""Interleaved A/B rounds (one process) of the two kernels for attention over at most 32 tokens (option attn_small: 0 persistent kernel, 1 one item per
wave) at the truncated text tower's shapes: CoOp per batch (500 prompts x 24 rows x 8 heads), zero-shot (1000 x 16), CoCoOp (3200 prompts x 24)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops

def dev_us(fn, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for n, l, h in [(500, 24, 8), (1000, 16, 8), (3200, 24, 8), (12800, 24, 8)]:
    qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
    res = {0: [], 1: []}
    for m in (0, 1):
        _lib.set_option("attn_small", m)
        dev_us(lambda: ops.attention(qkv, n, l, h, True), 5)
    for rnd in range(5):
        for m in (0, 1):
            _lib.set_option("attn_small", m)
            res[m].append(dev_us(lambda: ops.attention(qkv, n, l, h, True)))
    byts = 2.0 * n * l * 4 * 64 * h
    print(f"n={n} l={l} h={h} causal ({byts/1e6:.1f} MB): " + " | ".join(f"attn_small {m}: med {sorted(v)[2]:6.1f} us ({byts/sorted(v)[2]/1e6:5.2f} TB/s)" for m, v in res.items()), flush=True)
_lib.set_option("attn_small", 1)
