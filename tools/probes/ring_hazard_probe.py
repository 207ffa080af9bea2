#!/usr/bin/env python3
"""The LayerNorm check of tests/test_gpu_threads.py beside one attention kernel looping on a second stream, for the library CLIPMI_LIBRARY points at:
how many of N LayerNorm launches (one wave per row, 56 registers: it fits beside the hammer's waves) return a wrong row.  Hammers: the ring kernel at
257 and 577 tokens, the streaming kernel (attn_ring 0), the one-item-per-wave kernel (24 tokens, causal) and the persistent kernel on the same shape."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clip_calibration_amd import _lib, ops
N = int(os.environ.get("N", "200"))
g = torch.Generator().manual_seed(0)
M, K = 197 * 256, 768
x = torch.randn(M, K, generator=g).cuda()
gam, bet = torch.ones(K).cuda(), torch.zeros(K).cuda()
truth = torch.nn.functional.layer_norm(x.double(), (K,)).float()
HAMMERS = [("ring 257", (16, 257, 16, False), {"attn_ring": 1}), ("ring 577", (8, 577, 16, False), {"attn_ring": 1}), ("stream 257", (16, 257, 16, False), {"attn_ring": 0}),
           ("small 24 causal", (500, 24, 8, True), {"attn_small": 1}), ("persist 24 causal", (500, 24, 8, True), {"attn_small": 0})]
only = os.environ.get("ONLY")
out = []
for name, (n, l, h, causal), opts in HAMMERS:
    if only and only not in name:
        continue
    for k, v in opts.items():
        _lib.set_option(k, v)
    qkv = torch.randn(n * l, 3 * 64 * h, generator=g).half().cuda()
    stop, errors, launches = threading.Event(), [], [0]
    def hammer():
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                while not stop.is_set():
                    ops.attention(qkv, n, l, h, causal)
                    launches[0] += 1
                    if launches[0] % 100 == 0:
                        s.synchronize()
                s.synchronize()
        except Exception as e:
            errors.append(e)
    t = threading.Thread(target=hammer)
    t.start()
    bad = 0
    try:
        for _ in range(N):
            o = ops.layernorm(x, gam, bet)
            bad += int(bool(((o - truth).abs().amax(dim=1) > 1e-3).any()))
    finally:
        stop.set(); t.join()
    assert not errors, errors
    out.append(f"{name}: {bad} of {N}")
    _lib.set_option("attn_ring", 1); _lib.set_option("attn_small", 1)
print(f"{os.path.basename(os.environ.get('CLIPMI_LIBRARY', 'libclipmi.so')):26s} " + " | ".join(out), flush=True)
