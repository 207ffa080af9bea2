#!/usr/bin/env python3
"""Tuning aid: interleaved A/B rounds (one process, cdna_hip_programming.md rule 24) of the vision attention kernel variants
(option attn_loader: 0 persistent kernel, 1 loader wave + fragment reads pinned ahead by inline asm + full-line stores, 2 the same with
non-temporal stores (default)) at the image tower's shape, on random data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops

def dev_us(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

modes = [int(x) for x in os.environ.get("MODES", "0,1,2").split(",")]
for n, l, h in [(256, 197, 12), (256, 199, 12), (64, 197, 12)]:
    qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
    res = {m: [] for m in modes}
    for m in modes:
        _lib.set_option("attn_loader", m)
        dev_us(lambda: ops.attention(qkv, n, l, h, False), 10)
    for rnd in range(5):
        for m in modes:
            _lib.set_option("attn_loader", m)
            res[m].append(dev_us(lambda: ops.attention(qkv, n, l, h, False)))
    flop = 4.0 * n * h * l * l * 64
    print(f"n={n} l={l} h={h}: " + " | ".join(f"mode {m}: med {sorted(v)[len(v)//2]:6.1f} us min {min(v):6.1f} ({flop/sorted(v)[len(v)//2]/1e6:5.0f} TF)" for m, v in res.items()), flush=True)
_lib.set_option("attn_loader", 2)
