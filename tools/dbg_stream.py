import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
for (M, N, K) in [(3000, 520, 640), (3000, 512, 640), (3000, 520, 768), (256, 520, 640), (3000, 264, 640), (700, 776, 640)]:
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).half().cuda(); w = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    outs = {}
    for v in (1, 13):
        _lib.set_option("gemm_variant", v)
        outs[v] = ops.gemm_f16(a, w, bias, epilogue=_lib.EPI_BIAS, out_dtype=torch.float16).float()
    d = (outs[1] - outs[13]).abs()
    bad = (d > 1e-3).nonzero()
    print((M, N, K), "max diff", float(d.max()), "n bad", bad.shape[0])
    if bad.shape[0]:
        rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
        print("   bad rows", rows[:10].tolist(), "...", rows[-5:].tolist(), "count", rows.numel())
        print("   bad cols", cols[:10].tolist(), "...", cols[-5:].tolist(), "count", cols.numel())
        r, c = bad[0].tolist()
        print("   first bad", r, c, float(outs[1][r, c]), float(outs[13][r, c]))
