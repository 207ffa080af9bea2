#!/usr/bin/env python3
"""Tuning aid: time vs K at fixed M,N to separate the per-tile fixed cost (prologue+epilogue) from the main loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
from gemm_bench import timeit

M = 256 * 197
g = torch.Generator(device="cuda").manual_seed(0)
for name, n, epi, odt in [("fc/gelu f16out", 3072, _lib.EPI_BIAS_QUICKGELU, torch.float16), ("qkv/bias f16out", 2304, _lib.EPI_BIAS, torch.float16),
                          ("none f16out", 3072, _lib.EPI_NONE, torch.float16), ("res f32", 768, _lib.EPI_BIAS_RESIDUAL, torch.float32)]:
    for v in os.environ.get("VARIANTS", "1,8").split(","):
        _lib.set_option("gemm_variant", _lib.gemm_variant_id(v))
        row = [f"{name:16s} N={n} v{v}:"]
        for k in (64, 128, 256, 768, 1536, 3072):
            a = torch.randn(M, k, device="cuda", generator=g).half()
            w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).half()
            bias = torch.randn(n, device="cuda", generator=g)
            out = torch.empty(M, n, dtype=odt, device="cuda")
            res = out if epi == _lib.EPI_BIAS_RESIDUAL else None
            t = timeit(lambda: ops.gemm_f16(a, w, bias, res, epi, odt, out=out))
            row.append(f" K={k}: {t*1e3:6.1f}us")
        print("".join(row), flush=True)
