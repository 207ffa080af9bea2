#!/usr/bin/env python3
"""Diagnostic (never a benchmark): per-workgroup s_memrealtime stamps of the 2-stage GEMM kernel -> how long prologue,
main loop, epilogue issue and store drain take, and how synchronised the CUs are."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops

M = 256 * 197
CASES = [("qkv", 2304, 768, _lib.EPI_BIAS, torch.float16, "1"), ("fc", 3072, 768, _lib.EPI_BIAS_QUICKGELU, torch.float16, "1"),
         ("proj", 768, 3072, _lib.EPI_BIAS_RESIDUAL, torch.float32, "a"),
         ("qkv persistent", 2304, 768, _lib.EPI_BIAS, torch.float16, "b"), ("fc persistent", 3072, 768, _lib.EPI_BIAS_QUICKGELU, torch.float16, "b")]
for name, n, k, epi, odt, var in CASES:
    os.environ["CLIPMI_GEMM_VARIANT"] = var
    a = torch.randn(M, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    bias = torch.randn(n, device="cuda"); out = torch.empty(M, n, dtype=odt, device="cuda")
    res = out if epi == _lib.EPI_BIAS_RESIDUAL else None
    stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.gemm_f16(a, w, bias, res, epi, odt, out=out)
    os.environ["CLIPMI_GEMM_STAMPS_PTR"] = hex(stamps.data_ptr())
    ops.gemm_f16(a, w, bias, res, epi, odt, out=out)
    torch.cuda.synchronize()
    del os.environ["CLIPMI_GEMM_STAMPS_PTR"]
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] > 0]
    t = (s[:, :5] - s[:, 0].min()) / 100.0      # 100 MHz -> microseconds
    order = np.argsort(t[:, 0])
    t = t[order]
    if var == "b":   # persistent: stamps are per tile (virtual block id); column 5 = physical workgroup
        wg = s[order][:, 5]
        pro, main, epi_i = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        per_wg = {}
        for row, g in zip(t, wg):
            per_wg.setdefault(int(g), []).append(row)
        gaps = [b[0] - a[3] for rows in per_wg.values() for a, b in zip(sorted(rows, key=lambda r: r[0])[:-1], sorted(rows, key=lambda r: r[0])[1:])]
        print(f"{name}: {len(t)} tiles on {len(per_wg)} workgroups, span {t[:,3].max():.1f} us")
        print(f"   tile start -> first stage ready med {np.median(pro):5.2f} us  p90 {np.percentile(pro,90):5.2f}")
        print(f"   main loop med {np.median(main):5.2f} us  p90 {np.percentile(main,90):5.2f}")
        print(f"   prefetch + epilogue issue med {np.median(epi_i):5.2f} us  p90 {np.percentile(epi_i,90):5.2f}")
        print(f"   epilogue end -> next tile start med {np.median(gaps):5.2f} us")
        continue
    pro, main, epi_i, drain = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    print(f"{name}: {len(t)} workgroups, kernel span {t[:,4].max():.1f} us")
    print(f"   prologue  med {np.median(pro):5.2f} us  p90 {np.percentile(pro,90):5.2f}")
    print(f"   main loop med {np.median(main):5.2f} us  p90 {np.percentile(main,90):5.2f}")
    print(f"   epi issue med {np.median(epi_i):5.2f} us  p90 {np.percentile(epi_i,90):5.2f}")
    print(f"   drain     med {np.median(drain):5.2f} us  p90 {np.percentile(drain,90):5.2f}")
    first = t[:256]
    print(f"   first 256 WGs start spread {first[:,0].max()-first[:,0].min():.2f} us; their end spread {first[:,4].max()-first[:,4].min():.2f} us")
    starts = np.sort(t[:, 0])
    print("   start times of WG #256..#263 (second round):", np.round(starts[256:264], 1))
