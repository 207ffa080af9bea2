#!/usr/bin/env python3
"""Diagnostic (never a benchmark): per-workgroup s_memrealtime stamps of the GEMM kernels AS THE IMAGE TOWER LAUNCHES THEM
(LayerNorm fold, fp16 stream; clipmi_profile_block) -> how long prologue, main loop, epilogue issue and store drain take,
and how synchronised the CUs are.  Needs the tuning build of the library:

    make -C clip_calibration_amd/csrc tuning
    CLIPMI_LIBRARY=clip_calibration_amd/csrc/libclipmi_tuning.so python tools/gemm_stamps.py

The product build has no stamp code in its kernels and does not export clipmi_tuning_set_stamps."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn  # noqa: E402
from clip_calibration_amd.model import build_model  # noqa: E402

if not hasattr(_lib.lib, "clipmi_tuning_set_stamps"):
    raise SystemExit("libclipmi.so is the product build: `make -C clip_calibration_amd/csrc tuning` and set CLIPMI_LIBRARY")
_lib.lib.clipmi_tuning_set_stamps.argtypes = [ctypes.c_void_p]

B = int(os.environ.get("B", "256"))
WARM_S = float(os.environ.get("WARM_S", "0"))
model = build_model(dict(syn.synthetic_state_dict("ViT-B/16", seed=0)), None).cuda()
images = syn.synthetic_images(B, "ViT-B/16", seed=0, device="cuda")
with torch.no_grad():
    model.image_features_f32(images)          # real activations in the workspace
torch.cuda.synchronize()

# (name, step of the block, forced gemm_variant or None = one tile per workgroup); out-proj's row-range kernel: tools/rstream_stamps.py
CASES = [("in_proj", 0, None), ("out_proj", 2, 10), ("c_fc", 3, None), ("c_proj", 4, 10), ("in_proj stream", 0, 13), ("c_fc stream", 3, 13)]
if os.environ.get("CASES"):
    CASES = [c for c in CASES if c[0] in os.environ["CASES"].split(",")]
for name, step, variant in CASES:
    _lib.set_option("gemm_variant", -1 if variant is None else variant)
    _lib.set_option("gemm_stream", 0 if variant is None else 1)
    stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
    model.profile_block_ms(B, iters=2, only=step)
    if WARM_S > 0:   # steady-state clocks: the whole tower back to back for WARM_S seconds right before the stamped launch
        import time
        t_end = time.time() + WARM_S
        with torch.no_grad():
            while time.time() < t_end:
                for _ in range(10):
                    model.image_features_f32(images)
                torch.cuda.synchronize()
    _lib.lib.clipmi_tuning_set_stamps(stamps.data_ptr())
    ms = model.profile_block_ms(B, iters=1, only=step)[model.BLOCK_KERNELS[step]]   # warm-up launch + timed launch: the last one's stamps stay
    torch.cuda.synchronize()
    _lib.lib.clipmi_tuning_set_stamps(None)
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] > 0]
    bid = np.nonzero(stamps.cpu().numpy().reshape(-1, 8)[:, 0] > 0)[0]
    t = (s[:, :5] - s[:, 0].min()) / 100.0      # 100 MHz -> microseconds
    order = np.argsort(t[:, 0])
    t = t[order]
    # in-kernel shader clock over the main loop: d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back 6)
    dreal = (s[:, 2] - s[:, 1]).astype(np.float64)
    ok = (dreal > 0) & (s[:, 6] > 0) & (s[:, 7] > s[:, 6])
    if ok.any():
        clk = (s[ok, 7] - s[ok, 6]) / dreal[ok] * 100.0
        print(f"   in-kernel clock over the main loop: med {np.median(clk):.0f} MHz  p10 {np.percentile(clk,10):.0f}  p90 {np.percentile(clk,90):.0f}")
    if variant is None:   # one tile per workgroup: idle time of a CU between two workgroups (CU = (hardware id, XCD = blockIdx % 8))
        cu = {}
        for row, hw, b in zip(t, s[order][:, 5], bid[order]):
            cu.setdefault((int(hw), int(b) & 7), []).append(row)
        gaps = [b[0] - a[4] for rows in cu.values() for a, b in zip(rows[:-1], rows[1:])]
        busy = [sum(r[4] - r[0] for r in rows) for rows in cu.values()]
        if gaps:
            print(f"   {len(cu)} CUs; workgroups per CU {min(len(r) for r in cu.values())}..{max(len(r) for r in cu.values())}; "
                  f"end -> next start on the same CU med {np.median(gaps):5.2f} us  p90 {np.percentile(gaps,90):5.2f}; "
                  f"busy per CU med {np.median(busy):.1f} us; last end per CU p10 {np.percentile([r[-1][4] for r in cu.values()],10):.1f} "
                  f"med {np.median([r[-1][4] for r in cu.values()]):.1f} max {max(r[-1][4] for r in cu.values()):.1f}")
    if variant in (11, 13):   # persistent: stamps are per tile (virtual block id); column 5 = physical workgroup
        wg = s[order][:, 5]
        pro, main, epi_i = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        per_wg = {}
        for row, g in zip(t, wg):
            per_wg.setdefault(int(g), []).append(row)
        gaps = [b[0] - a[3] for rows in per_wg.values() for a, b in zip(sorted(rows, key=lambda r: r[0])[:-1], sorted(rows, key=lambda r: r[0])[1:])]
        print(f"{name}: {len(t)} tiles on {len(per_wg)} workgroups, span {t[:,3].max():.1f} us (hipEvents {ms*1e3:.1f} us)")
        print(f"   tile start -> first stage ready med {np.median(pro):5.2f} us  p90 {np.percentile(pro,90):5.2f}")
        print(f"   main loop med {np.median(main):5.2f} us  p90 {np.percentile(main,90):5.2f}")
        print(f"   prefetch + epilogue issue med {np.median(epi_i):5.2f} us  p90 {np.percentile(epi_i,90):5.2f}")
        print(f"   epilogue end -> next tile start med {np.median(gaps):5.2f} us")
        continue
    raw = stamps.cpu().numpy().reshape(-1, 8)
    parts = raw[4096:4096 + 4096]
    parts = parts[parts[:, 0] > 0]
    if len(parts):   # gemm_pp_kernel's part timers (shader cycles summed over the K loop; group 0 = wave 0, group 1 = wave 4)
        nk = {"out_proj": 12, "c_proj": 48}.get(name.split()[0], 12) * 4
        for g in (0, 1):
            p4 = parts[:, 4 * g:4 * g + 4].astype(np.float64) / nk
            print(f"   group {g} per phase (cycles): load part {np.median(p4[:,0]):.0f}  wait at barrier {np.median(p4[:,1]):.0f}  "
                  f"compute part {np.median(p4[:,2]):.0f}  wait at barrier {np.median(p4[:,3]):.0f}")
    pro, main, epi_i, drain = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    print(f"{name}: {len(t)} workgroups, kernel span {t[:,4].max():.1f} us (hipEvents {ms*1e3:.1f} us)")
    print(f"   prologue  med {np.median(pro):5.2f} us  p90 {np.percentile(pro,90):5.2f}")
    print(f"   main loop med {np.median(main):5.2f} us  p90 {np.percentile(main,90):5.2f}")
    print(f"   epi issue med {np.median(epi_i):5.2f} us  p90 {np.percentile(epi_i,90):5.2f}")
    print(f"   drain     med {np.median(drain):5.2f} us  p90 {np.percentile(drain,90):5.2f}")
    n1 = min(256, len(t))
    first = t[:n1]
    print(f"   first {n1} WGs start spread {first[:,0].max()-first[:,0].min():.2f} us; their end spread {first[:,4].max()-first[:,4].min():.2f} us")
    starts = np.sort(t[:, 0])
    if len(starts) > 264:
        print("   start times of WG #256..#263 (second round):", np.round(starts[256:264], 1))
_lib.set_option("gemm_variant", -1)
