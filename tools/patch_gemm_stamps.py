#!/usr/bin/env python3
"""Diagnostic (tuning build): phase stamps of the patch-embedding GEMM (gemm_pp_kernel<320 x 256, EPI_PATCH_POS, IM2COL>) launched through
clipmi_patch_embed on 256 fp32 images -- prologue, main loop, epilogue issue, store drain per tile, and the two rounds of tiles.
    CLIPMI_LIBRARY=clip_calibration_amd/csrc/libclipmi_tuning.so python tools/patch_gemm_stamps.py"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clip_calibration_amd import _lib
from clip_calibration_amd._lib import lib, check, F16, F32
lib.clipmi_tuning_set_stamps.argtypes = [ctypes.c_void_p]
B, R, P, D = 256, 224, 16, 768
G = R // P; L = 1 + G * G
img = torch.randn(B, 3, R, R, device="cuda")
w = (torch.randn(D, 3 * P * P, device="cuda") * 0.03).half()
x0 = torch.empty(B * L, D, dtype=torch.float16, device="cuda")
scratch = torch.empty(lib.clipmi_patch_embed_scratch_bytes(B, R, F32), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run():
    check(lib.clipmi_patch_embed(img.data_ptr(), F32, scratch.data_ptr(), w.data_ptr(), 3 * P * P, None, x0.data_ptr(), F16, B, R, P, D, L, st), "pe")
for _ in range(5): run()
torch.cuda.synchronize()
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
lib.clipmi_tuning_set_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize()
lib.clipmi_tuning_set_stamps(None)
s = stamps.cpu().numpy().reshape(-1, 8); s = s[s[:, 0] > 0]
t = (s[:, :5] - s[:, 0].min()) / 100.0
print(len(s), "tiles; span", t[:, 4].max(), "us")
print("prologue med %.2f | main loop med %.2f | epilogue issue med %.2f | drain med %.2f us" % tuple(np.median(t[:, i + 1] - t[:, i]) for i in range(4)))
first = t[t[:, 0] < 5]; second = t[t[:, 0] >= 5]
print("first round: %d tiles, end med %.1f; second round: %d tiles, start med %.1f end med %.1f max %.1f" % (len(first), np.median(first[:, 4]), len(second), np.median(second[:, 0]), np.median(second[:, 4]), second[:, 4].max()))
dreal = (s[:, 2] - s[:, 1]).astype(float); ok = (dreal > 0) & (s[:, 6] > 0)
print("clock over main loop med %.0f MHz" % np.median((s[ok, 7] - s[ok, 6]) / dreal[ok] * 100))
