#!/usr/bin/env python3
"""Fifty passes of the CoOp text tower as BASELINE configs[2] runs it per batch (500 classes, n_ctx 16, fp16 stream, dead rows eliminated; CLASSES /
ROWS=77 override): the program rocprofv3 --kernel-trace --stats wraps to see where a truncated text tower spends its time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import _lib, synthetic as syn
from clip_calibration_amd.model import build_model
from clip_calibration_amd.trainers import CoOpCLIP
C = int(os.environ.get("CLASSES", "500"))
model = build_model(syn.synthetic_state_dict("ViT-B/16"), {"trainer": "CoOp"}).cuda()
model.text_dead_row_elimination = os.environ.get("ROWS", "") != "77"
coop = CoOpCLIP(model, syn.synthetic_token_ids(C, "ViT-B/16", seed=11, n_ctx_placeholders=16), n_ctx=16, logit_scale=1.0, seed=3, cache_text_features=False)
with torch.no_grad():
    for _ in range(50):
        f = coop.text_features()
torch.cuda.synchronize()
print("rows", model.live_rows(coop.tokenized_prompts), "finite", bool(torch.isfinite(f).all()))
