#!/usr/bin/env python3
"""Twenty launches of the long-sequence attention at one shape (SHAPE = n,l,h; default the ViT-L/14@336 per-rank shape 64,577,16): the program
rocprofv3 wraps for tools/measure_attn_sq.sh (counters) and for a kernel-trace of the kernel alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
n, l, h = [int(x) for x in os.environ.get("SHAPE", "64,577,16").split(",")]
_lib.set_option("attn_ring", int(os.environ.get("ATTN_RING", "1")))
qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
for _ in range(int(os.environ.get("LAUNCHES", "20"))):
    ops.attention(qkv, n, l, h, False)
torch.cuda.synchronize()
