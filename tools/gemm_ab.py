#!/usr/bin/env python3
"""Tuning aid: interleaved A/B of GEMM tile variants in ONE process (cdna_hip_programming.md §5.4 rule 24):
ROUNDS x variants x shapes, 5 launches per cell, reports median and min per variant."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops

B = int(os.environ.get("B", "256")); M = B * 197
SHAPES = [("qkv", M, 2304, 768, _lib.EPI_BIAS, torch.float16), ("out", M, 768, 768, _lib.EPI_BIAS_RESIDUAL, torch.float32),
          ("fc", M, 3072, 768, _lib.EPI_BIAS_QUICKGELU, torch.float16), ("proj", M, 768, 3072, _lib.EPI_BIAS_RESIDUAL, torch.float32)]
VARIANTS = os.environ.get("VARIANTS", "0,1,2,9").split(",")
ROUNDS = int(os.environ.get("ROUNDS", "12"))
g = torch.Generator(device="cuda").manual_seed(0)
for name, m, n, k, epi, odt in SHAPES:
    a = torch.randn(m, k, device="cuda", generator=g).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).half()
    bias = torch.randn(n, device="cuda", generator=g) * 0.1
    out = torch.empty(m, n, dtype=odt, device="cuda")
    res = out if epi == _lib.EPI_BIAS_RESIDUAL else None
    times = {v: [] for v in VARIANTS}
    for r in range(ROUNDS + 1):
        for v in VARIANTS:
            _lib.set_option("gemm_variant", _lib.gemm_variant_id(v))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_f16(a, w, bias, res, epi, odt, out=out)
            e1.record(); torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    flop = 2.0 * m * n * k
    print(f"{name:5s}", " | ".join(f"v{v}: med {statistics.median(t):6.1f} us ({flop/statistics.median(t)/1e6:6.1f} TF) min {min(t):6.1f}" for v, t in times.items()), flush=True)
