#!/usr/bin/env python3
"""Tuning aid: does the MLP pair (c_fc with QuickGELU -> c_proj with residual) get cheaper when it is run in row chunks, so that
a chunk of the [M, 4D] hidden activation (310 MB at batch 256) is still in the 256 MB Infinity Cache when c_proj reads it?
Interleaved arms, op-level GEMMs (no LayerNorm fold, fp32 residual): the question is only the hidden activation's traffic."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
M, D = int(os.environ.get("M", "50432")), 768
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, D, generator=g) * 0.5).half().cuda()
w1 = (torch.randn(4 * D, D, generator=g) * 0.03).half().cuda(); b1 = torch.randn(4 * D, generator=g).cuda() * 0.1
w2 = (torch.randn(D, 4 * D, generator=g) * 0.02).half().cuda(); b2 = torch.randn(D, generator=g).cuda() * 0.1
res = torch.randn(M, D, generator=g).cuda()
hid = torch.empty(M, 4 * D, dtype=torch.float16, device="cuda")
out = torch.empty(M, D, dtype=torch.float32, device="cuda")
ARMS = [int(c) for c in os.environ.get("CHUNKS", "1,2,3,4,6").split(",")]

def run(chunks):
    step = ((M + chunks - 1) // chunks + 255) // 256 * 256      # whole 256-row tiles per chunk
    for r0 in range(0, M, step):
        r1 = min(M, r0 + step)
        ops.gemm_f16(x[r0:r1], w1, b1, epilogue=_lib.EPI_BIAS_QUICKGELU, out=hid[r0:r1])
        ops.gemm_f16(hid[r0:r1], w2, b2, residual=res[r0:r1], epilogue=_lib.EPI_BIAS_RESIDUAL, out_dtype=torch.float32, out=out[r0:r1])

times = {c: [] for c in ARMS}
for rnd in range(int(os.environ.get("ROUNDS", "6")) + 1):
    for c in ARMS:
        run(c); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(12):
            run(c)
        e1.record(); torch.cuda.synchronize()
        if rnd:
            times[c].append(e0.elapsed_time(e1) / 12 * 1e3)
for c in ARMS:
    print(f"chunks {c}: c_fc + c_proj over {M} rows  med {statistics.median(times[c]):7.1f} us  (min {min(times[c]):7.1f})", flush=True)
