#!/usr/bin/env python3
"""Tuning aid: device time of the tail (fused launch vs separate launches), queued behind a long kernel so that the host never
starves the stream.  Also the fused kernel without its row pass (logits only) to see the matmul phase alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops

def dev_time(fn, n=40):
    fn(); torch.cuda.synchronize()
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(6):
        big.add_(1.0)            # ~1 ms each of GPU work ahead of the measured launches
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B, C, E in [(256, 1000, 512), (512, 1000, 512), (1024, 1000, 512), (2048, 1000, 512), (128, 199, 512), (256, 500, 512), (64, 1000, 768)]:
    img = torch.randn(B, E, device="cuda") * 3
    txt = ops.l2_normalize(torch.randn(C, E, device="cuda"))
    dac = torch.rand(C, device="cuda") + 0.5
    labels = torch.randint(0, C, (B,), device="cuda")
    bins = torch.zeros(33, dtype=torch.float64, device="cuda")
    byts = 4.0 * (B * E + C * E) + 4.0 * B * C + 16.0 * B
    row = [f"B={B} C={C}"]
    for name, unf, kw in [("fused", 0, {}), ("fused+dac", 0, {"dac_conf": dac}), ("fused logits only", 0, {"want_conf_pred": False, "labels": None, "bins": None}),
                          ("separate launches", 1, {})]:
        _lib.set_option("tail_unfused", unf)
        args = dict(dac_conf=None, want_conf_pred=True, normalize=True, labels=labels, bins=bins, n_bins=10); args.update(kw)
        us = dev_time(lambda: ops.fused_tail(img, txt, 100.0, **args))
        row.append(f"{name} {us:6.1f} us ({byts/us/1e3:6.1f} GB/s)")
    _lib.set_option("tail_unfused", 0)
    print(" | ".join(row), flush=True)
