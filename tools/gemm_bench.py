#!/usr/bin/env python3
"""Tuning aid (not part of the product): times the HIP GEMM tile variants on the tower's GEMM shapes with random data,
checks them against torch.matmul, and prints hipBLASLt's time on the same problem as a known-good reference point."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops  # noqa: E402

B = int(os.environ.get("B", "256"))
M = B * 197
SHAPES = [("qkv", M, 2304, 768, _lib.EPI_BIAS, torch.float16),
          ("out", M, 768, 768, _lib.EPI_BIAS_RESIDUAL, torch.float32),
          ("fc", M, 3072, 768, _lib.EPI_BIAS_QUICKGELU, torch.float16),
          ("proj", M, 768, 3072, _lib.EPI_BIAS_RESIDUAL, torch.float32)]
if os.environ.get("TOWER", "vision") == "text":      # ViT-B/16's text tower: C prompts x 77 tokens, width 512
    M = int(os.environ.get("C", "4000")) * 77
    SHAPES = [("qkv", M, 1536, 512, _lib.EPI_BIAS, torch.float16),
              ("out", M, 512, 512, _lib.EPI_BIAS_RESIDUAL, torch.float32),
              ("fc", M, 2048, 512, _lib.EPI_BIAS_QUICKGELU, torch.float16),
              ("proj", M, 512, 2048, _lib.EPI_BIAS_RESIDUAL, torch.float32)]
VARIANTS = os.environ.get("VARIANTS", "0,1,2,3,4,5,6,7,8").split(",")


ITERS = int(os.environ.get("ITERS", "20"))


def timeit(fn, iters=None):
    iters = iters or ITERS
    for _ in range(min(3, iters)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, m, n, k, epi, odt in SHAPES:
        a = torch.randn(m, k, device="cuda", generator=g).half()
        w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).half()
        if os.environ.get("ZERO") == "1":      # power / clock probe: the same instruction stream on all-zero operands
            a.zero_(); w.zero_()
        bias = torch.randn(n, device="cuda", generator=g) * 0.1
        res = torch.randn(m, n, device="cuda", generator=g) if epi == _lib.EPI_BIAS_RESIDUAL else None
        flop = 2.0 * m * n * k
        ref = None
        t_blas = timeit(lambda: torch.matmul(a, w.t()))
        line = [f"{name:5s} M={m} N={n} K={k}  hipBLASLt(no epilogue) {t_blas*1e3:7.1f} us {flop/t_blas/1e9:7.1f} TF |"]
        ref = (a[:512].float() @ w.float().t()) + bias
        for v in VARIANTS:
            _lib.set_option("gemm_variant", _lib.gemm_variant_id(v))
            out = torch.empty(m, n, dtype=odt, device="cuda")
            r = res.clone() if res is not None else None
            got = ops.gemm_f16(a, w, bias, r, epi, odt, out=out if r is None else r)
            chk = got[:512].float()
            if epi == _lib.EPI_BIAS_QUICKGELU:
                want = ref * torch.sigmoid(1.702 * ref)
            elif epi == _lib.EPI_BIAS_RESIDUAL:
                want = ref + res[:512]
            else:
                want = ref
            err = (chk - want).abs().max().item() / (want.abs().max().item() + 1e-9)
            buf = r if r is not None else out
            t = timeit(lambda: ops.gemm_f16(a, w, bias, buf if r is not None else None, epi, odt, out=buf))
            line.append(f" v{v}: {t*1e3:7.1f} us {flop/t/1e9:7.1f} TF err {err:.1e} |")
        print("".join(line), flush=True)


if __name__ == "__main__":
    main()
