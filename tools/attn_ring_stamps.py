#!/usr/bin/env python3
"""Diagnostic (tuning build, CLIPMI_LIBRARY=.../libclipmi_tuning.so): where a step of the ring attention kernel spends its cycles.
Stamps (s_memtime, shader cycles, lane 0 of every wave, workgroups 0-7, steps 0-63).  Compute waves 0-10: 0 loop top | 2 past the block's barrier |
3 S tiles of the last group issued | 4 its maximum + rescale done | 5 block computed | 6 step end (pass end: merge + store).  Loader (wave 11): 0 loop
top | 1 block g landed | 2 past the barrier | 7 DMA of block g + 2 (and the next Q tiles) issued."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, ops
assert hasattr(_lib.lib, "clipmi_tuning_set_stamps"), "needs CLIPMI_LIBRARY=.../libclipmi_tuning.so"
_lib.lib.clipmi_tuning_set_stamps.argtypes = [ctypes.c_void_p]
n, l, h = [int(x) for x in os.environ.get("SHAPE", "64,577,16").split(",")]
qkv = torch.randn(n * l, 3 * 64 * h, device="cuda").half()
for _ in range(3):
    ops.attention(qkv, n, l, h, False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention(qkv, n, l, h, False)
e1.record()
torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('CLIPMI_LIBRARY', 'libclipmi.so'))}: n={n} l={l} h={h}: {e0.elapsed_time(e1) * 50:.1f} us per launch (20 launches, no stamps)")
NW = 12                                                                # 11 compute waves + the loader (attention.hip RNW + 1: the kernel indexes the buffer with it)
stamps = torch.zeros(8 * 64 * NW * 8, dtype=torch.int64, device="cuda")
_lib.lib.clipmi_tuning_set_stamps(stamps.data_ptr())
ops.attention(qkv, n, l, h, False)
torch.cuda.synchronize()
_lib.lib.clipmi_tuning_set_stamps(None)
s = stamps.cpu().numpy().reshape(8, 64, NW, 8).astype(np.float64)     # [wg][step][wave][stamp]
nb = (l + 127) // 128
npass = ((l + 31) // 32 + NW - 2) // (NW - 1)
steps = min(64, (n * h // 256) * npass * nb)
s = s[:, :steps]
CW = NW - 1
print(f"SIMD of compute waves 0-{CW - 1} (HW_ID.SIMD_ID), workgroups 0-3:", [[int(s[wg, 0, w, 1]) for w in range(CW)] for wg in range(4)])
print(f"{steps} steps per workgroup ({npass} passes x {nb} blocks per item); span of workgroup 0: {(s[0, :, :CW, 6].max() - s[0, 0, :CW, 0].min()):.0f} cycles")
for kind, sel in (("full-pass steps (not the last block)", lambda p, b: p < npass - 1 and b < nb - 1), ("full-pass LAST block", lambda p, b: p < npass - 1 and b == nb - 1),
                  ("split-pass steps (not the last block)", lambda p, b: p == npass - 1 and b < nb - 1), ("split-pass LAST block", lambda p, b: p == npass - 1 and b == nb - 1)):
    idx = [g for g in range(steps) if sel((g // nb) % npass, g % nb)]
    if not idx:
        continue
    t = s[:, idx]                                                   # [wg][step][wave][8]
    step_len = t[:, :, :CW, 6] - t[:, :, :CW, 0]
    print(f"--- {kind}: {len(idx)} steps x 8 workgroups; mean step {step_len.mean():.0f} cycles")
    for w in range(CW):
        x = t[:, :, w, :]
        ok = x[..., 3] > 0                                          # waves that computed in this step
        bar = (x[..., 2] - x[..., 0]).mean()
        if ok.any():
            comp = (x[..., 5] - x[..., 2])[ok].mean()
            lastgrp = (x[..., 5] - x[..., 3])[ok].mean()
            mx = (x[..., 4] - x[..., 3])[ok].mean()
        else:
            comp = lastgrp = mx = float("nan")
        print(f"   wave {w}: barrier wait {bar:6.0f} | compute {comp:6.0f} (last group: S issued -> max done {mx:5.0f}, S issued -> group done {lastgrp:6.0f}) | "
              f"pass end {(x[..., 6] - x[..., 5]).mean():6.0f}")
    x = t[:, :, CW, :]
    print(f"   loader: wait landing {(x[..., 1] - x[..., 0]).mean():6.0f} | barrier {(x[..., 2] - x[..., 1]).mean():6.0f} | issue {(x[..., 7] - x[..., 2]).mean():6.0f}")
