#!/usr/bin/env python3
"""Diagnostic (tuning build, CLIPMI_LIBRARY=.../libclipmi_tuning.so): where a round of the fused in-projection + attention kernel
(clipmi_qkv_attention, attention.hip qkv_attention_kernel) spends its time.  Stamps per workgroup and round, wave 0 (a query wave) and
wave 7 (the loader): 0 round start | 1 attention / loader work issued | 2 ... complete | 3 past the barrier: GEMM phase starts | 4 K loop done |
5 past the barrier | 6 epilogue written."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
assert hasattr(_lib.lib, "clipmi_tuning_set_stamps"), "needs CLIPMI_LIBRARY=.../libclipmi_tuning.so"
_lib.lib.clipmi_tuning_set_stamps.argtypes = [ctypes.c_void_p]
n, l, h = 256, 197, 12
D = 64 * h
x = torch.randn(n * l, D, device="cuda").half()
w = (torch.randn(3 * D, D, device="cuda") * D ** -0.5).half()
b = torch.randn(3 * D, device="cuda") * 0.2
for _ in range(3):
    ops.qkv_attention(x, w, b, n, l, h)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    ops.qkv_attention(x, w, b, n, l, h)
e1.record()
torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('CLIPMI_LIBRARY', 'libclipmi.so'))}: fused launch {e0.elapsed_time(e1) / 30 * 1e3:.1f} us (30 launches, stamps off)")
stamps = torch.zeros(256 * 16 * 16, dtype=torch.int64, device="cuda")
_lib.lib.clipmi_tuning_set_stamps(stamps.data_ptr())
ops.qkv_attention(x, w, b, n, l, h)
torch.cuda.synchronize()
_lib.lib.clipmi_tuning_set_stamps(None)
s = stamps.cpu().numpy().reshape(256, 16, 2, 8).astype(np.float64) / 100.0    # us; [workgroup][round][wave 0 | wave 7][stamp]
mid = s[:, 2:11]                                                                  # rounds with a current AND a next item
q, ld = mid[:, :, 0], mid[:, :, 1]
print(f"kernel span {s[s > 0].max() - s[s > 0].min():.1f} us; per round (median over workgroups x rounds 2..10):")
print(f"  wave 0: attention + store issued {np.median(q[..., 1] - q[..., 0]):5.2f} | LDS drained {np.median(q[..., 2] - q[..., 1]):5.2f} | wait at barrier {np.median(q[..., 3] - q[..., 2]):5.2f} |"
      f" K loop {np.median(q[..., 4] - q[..., 3]):5.2f} | barrier {np.median(q[..., 5] - q[..., 4]):5.2f} | epilogue {np.median(q[..., 6] - q[..., 5]):5.2f} us")
print(f"  wave 7: loader issue {np.median(ld[..., 1] - ld[..., 0]):5.2f} | landing wait {np.median(ld[..., 2] - ld[..., 1]):5.2f} | wait at barrier {np.median(ld[..., 3] - ld[..., 2]):5.2f} |"
      f" K loop {np.median(ld[..., 4] - ld[..., 3]):5.2f} | barrier {np.median(ld[..., 5] - ld[..., 4]):5.2f} | epilogue {np.median(ld[..., 6] - ld[..., 5]):5.2f} us")
rnd = np.median(s[:, 3:11, 0, 0] - s[:, 2:10, 0, 0])
print(f"  round (start to start) {rnd:5.2f} us x 12 rounds = {12 * rnd:.1f} us")
