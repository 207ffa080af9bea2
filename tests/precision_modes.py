#!/usr/bin/env python3
"""Checker script (lives under tests/ because it uses the oracle): cosine-logit error of the towers against the CPU oracle under the residual-stream precision modes
(CLIPMI_RESIDUAL_F16 unset / v / t / 1), ViT-B/16 geometry, zero-shot and CoOp text sides."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
from clip_calibration_amd import synthetic as syn
from clip_calibration_amd.model import build_model
from clip_calibration_amd.trainers import CoOpCLIP
from oracle import clip_oracle as orc    # checker only

G = os.environ.get("GEOM", "ViT-B/16")
B, C, n_ctx = int(os.environ.get("B", "8")), int(os.environ.get("C", "24")), 16
sd = syn.synthetic_state_dict(G, seed=0)
images = syn.synthetic_images(B, G, seed=7)
ids_zs = syn.synthetic_token_ids(C, G, seed=11)
ids_cp = syn.synthetic_token_ids(C, G, seed=11, n_ctx_placeholders=n_ctx)
model = build_model(dict(sd), None).cuda()
coop = CoOpCLIP(model, ids_cp, n_ctx=n_ctx, seed=3)
ctx = coop.prompt_learner.ctx.detach().float().cpu()
with torch.no_grad():
    ri = orc.l2_normalize(orc.encode_image(sd, images)).numpy()
    rz = orc.l2_normalize(orc.encode_text(sd, ids_zs)).numpy()
    rc = orc.l2_normalize(orc.text_encoder(sd, orc.coop_prompts(sd, ids_cp, ctx), ids_cp)).numpy()
n = lambda a: a / np.linalg.norm(a, axis=1, keepdims=True)
from clip_calibration_amd import _lib
for mode, value in (("0", 0), ("v", 2), ("t", 3), ("1", 1)):
    _lib.set_option("residual_f16", value)
    coop._cache = None; coop._cache_key = None
    with torch.no_grad():
        gi = n(model.image_features_f32(images.cuda()).cpu().numpy())
        gz = n(model.text_features_f32(ids_zs.cuda()).cpu().numpy())
        gc = coop.text_features().cpu().numpy()
    print(f"mode {mode}: image-side |d cos| {np.abs(gi @ rz.T - ri @ rz.T).max():.2e}   zero-shot text-side {np.abs(ri @ gz.T - ri @ rz.T).max():.2e}   "
          f"CoOp text-side {np.abs(ri @ gc.T - ri @ rc.T).max():.2e}   both (zs) {np.abs(gi @ gz.T - ri @ rz.T).max():.2e}   both (CoOp) {np.abs(gi @ gc.T - ri @ rc.T).max():.2e}",
          flush=True)
