#!/usr/bin/env python3
"""Interleaved A/B rounds (one process, cdna_hip_programming.md rule 24) of the two kernels for non-causal attention over more than 224 tokens:
option attn_ring 0 = the round-1 streaming kernel, 1 = the ring kernel (round 5), at the ViT-L/14 shapes of BASELINE configs[4], on random data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops

def dev_us(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

modes = [int(x) for x in os.environ.get("MODES", "0,1").split(",")]
shapes = [(64, 577, 16), (128, 257, 16), (32, 577, 16), (256, 257, 16)]
for n, l, h in shapes:
    qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
    res = {m: [] for m in modes}
    for m in modes:
        _lib.set_option("attn_ring", m)
        dev_us(lambda: ops.attention(qkv, n, l, h, False), 5)
    for rnd in range(5):
        for m in modes:
            _lib.set_option("attn_ring", m)
            res[m].append(dev_us(lambda: ops.attention(qkv, n, l, h, False)))
    flop = 4.0 * n * h * l * l * 64
    byts = 2.0 * n * l * 4 * 64 * h
    print(f"n={n} l={l} h={h} ({flop/1e9:.1f} GFLOP, {byts/1e6:.0f} MB): " + " | ".join(
        f"attn_ring {m}: med {sorted(v)[len(v)//2]:6.1f} us min {min(v):6.1f} ({flop/sorted(v)[len(v)//2]/1e6:5.0f} TF)" for m, v in res.items()), flush=True)
_lib.set_option("attn_ring", 1)
