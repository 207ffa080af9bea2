#!/usr/bin/env python3
"""Diagnostic (tuning build only, never a benchmark): ablations of the streamed-epilogue GEMM kernel.  knob bits: 1 no held-slice
stores in the K loop, 2 no LDS-DMA pieces in the K loop (the stages keep their first contents), 4 no MFMAs, 8 no conversion of the
accumulators (timing only: results are wrong with any bit set).
    make -C clip_calibration_amd/csrc tuning && CLIPMI_LIBRARY=.../libclipmi_tuning.so python tools/stream_ablate.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model
assert hasattr(_lib.lib, "clipmi_tuning_set_knob"), "needs the tuning build (CLIPMI_LIBRARY)"
B = 256
model = build_model(dict(syn.synthetic_state_dict("ViT-B/16", seed=0)), None).cuda()
images = syn.synthetic_images(B, "ViT-B/16", seed=0, device="cuda")
with torch.no_grad():
    model.image_features_f32(images)
res = {}
for rnd in range(4):
    for stream, knob in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (1, 4), (1, 7), (1, 8), (1, 15)):
        _lib.set_option("gemm_stream", stream)
        _lib.lib.clipmi_tuning_set_knob(knob)
        with torch.no_grad():
            model.image_features_f32(images)
        ms = model.profile_block_ms(B, iters=6)
        if rnd:
            res.setdefault((stream, knob), []).append((ms["in_proj"] * 1e3, ms["c_fc"] * 1e3))
_lib.lib.clipmi_tuning_set_knob(0)
for (stream, knob), v in res.items():
    print(f"stream={stream} knob={knob:2d}: in_proj {statistics.median(x[0] for x in v):6.1f} us   c_fc {statistics.median(x[1] for x in v):6.1f} us", flush=True)
