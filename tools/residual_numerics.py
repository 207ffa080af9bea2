#!/usr/bin/env python3
"""Diagnostic: which of the residual-GEMM kernels is closer to the exact sum?  x + a @ w^T + bias in float64 on the host against the
fp16 outputs of gemm_variant 16 (row ranges, residual preloaded into the accumulators) and 10 (320 x 256 tiles, residual added last)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
torch.manual_seed(0)
for M, N, K, rs in ((50432, 768, 768, 1.0), (50432, 768, 3072, 1.0), (50432, 768, 768, 30.0)):
    a = (torch.randn(M, K) * 0.5).half()
    w = (torch.randn(N, K) * K ** -0.5).half()
    bias = torch.randn(N) * 0.1
    x0 = (torch.randn(M, N) * rs).half()
    exact = x0.double() + a.double() @ w.double().t() + bias.double()
    ref16 = exact.float().half()
    out = {}
    for v in (16, 10):
        _lib.set_option("gemm_variant", v)
        x = x0.clone().cuda()
        ops.gemm_residual_f16(a.cuda(), w.cuda(), bias.cuda(), x)
        out[v] = x.cpu()
    _lib.set_option("gemm_variant", -1)
    for v in (16, 10):
        d = (out[v].double() - exact)
        hu = (ref16.double() - exact)          # the unavoidable rounding error
        print(f"M={M} N={N} K={K} residual sigma {rs}: variant {v}: mean signed err {d.mean():+.3e} (ideal rounding {hu.mean():+.3e}), rms {d.pow(2).mean().sqrt():.3e} (ideal {hu.pow(2).mean().sqrt():.3e}), "
              f"mismatch vs correctly rounded {float((out[v] != ref16).double().mean()):.2e}, max |err| / |exact| on |exact| > 0.1: {float((d.abs() / exact.abs())[exact.abs() > 0.1].max()):.2e}")
