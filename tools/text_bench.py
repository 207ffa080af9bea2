#!/usr/bin/env python3
"""Tuning aid: text-tower throughput (prompts/s, TFLOP/s at 5.960 GFLOP/prompt for ViT-B/16's text tower) vs number of
prompts, and a CoCoOp step (B images x C classes -> B*C prompts) -- the f-4 workload whose cost is the TEXT tower."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model
from clip_calibration_amd.trainers import CoCoOpCLIP

G = os.environ.get("GEOM", "ViT-B/16")
sd = syn.synthetic_state_dict(G)
model = build_model(dict(sd), None).cuda()
fpp = syn.flops_per_prompt(G)

def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

for C in [int(c) for c in os.environ.get("CS", "100,500,1000,2000,4000,8000").split(",")]:
    ids = syn.synthetic_token_ids(C, G, seed=1).cuda()
    for what, flags in (("fp32 stream (default)", _lib.CALL_DEFAULT), ("fp16 stream (per-call flag)", _lib.CALL_STREAM_F16)):
        dt = timed(lambda: model.text_features_f32(ids, flags=flags), max(2, 4000 // C))
        rows = model.live_rows(ids)       # dead-row elimination (round 5): flop is credited for the token rows that are computed
        print(f"encode_text C={C:5d} {what:28s}: {dt*1e3:8.2f} ms  {C/dt:9.0f} prompts/s  {rows:2d} of {model.context_length} rows  "
              f"{C*fpp*rows/model.context_length/dt/1e12:7.1f} TFLOP/s on computed rows", flush=True)

if os.environ.get("COCOOP", "1") == "1":
    for B, C, per_call in ((32, 100, 3200), (32, 100, 800), (16, 1000, 4000), (16, 1000, 8000)):
        ids = syn.synthetic_token_ids(C, G, seed=2, n_ctx_placeholders=4).cuda()
        img = syn.synthetic_images(B, G, device="cuda")
        co = CoCoOpCLIP(model, ids, n_ctx=4, prompts_per_call=per_call)
        dt = timed(lambda: co(img, want_conf_pred=True), 3)
        rows = model.live_rows(co.tokenized_prompts)
        print(f"CoCoOp B={B} C={C} prompts/call={per_call}: {dt*1e3:8.1f} ms/step  {B/dt:7.1f} img/s  {B*C/dt:9.0f} prompts/s  {rows} of {model.context_length} rows  "
              f"{(B*C*fpp*rows/model.context_length + B*syn.flops_per_image(G))/dt/1e12:7.1f} TFLOP/s on computed rows", flush=True)
