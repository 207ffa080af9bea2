#!/usr/bin/env python3
"""Tuning aid: attention time per (sequence, head) item against the batch size -- is the kernel memory- or issue-bound?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import ops
from gemm_bench import timeit
for n in (22, 43, 64, 128, 256, 512):
    l, h = 197, 12
    qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
    t = timeit(lambda: ops.attention(qkv, n, l, h, False), 30)
    items = n * h
    print(f"B={n:4d}: {t*1e3:7.1f} us  = {t*1e6/items*256/1e3:6.2f} us per item per CU-slot ({items/256:.1f} items per CU), "
          f"{(qkv.numel()+n*l*64*h)*2/t/1e9:6.2f} TB/s", flush=True)
