#!/usr/bin/env python3
"""Measurement: the reference's CoOp TextEncoder statements (trainers/classification/coop.py:56-67) on the swapped ``build_model`` (INTEGRATION
level 1), per batch, with and without the opt-in row hint ``clip_model.transformer.live_rows = tokenized_prompts``."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_calibration_amd import synthetic as syn
from clip_calibration_amd.model import build_model

G = os.environ.get("GEOM", "ViT-B/16")
model = build_model(dict(syn.synthetic_state_dict(G)), None).cuda()
for Cn in (100, 1000):
    ids = syn.synthetic_token_ids(Cn, G, seed=2, n_ctx_placeholders=16).cuda()
    with torch.no_grad():
        prompts = model.token_embedding(ids).type(model.dtype)

        def text_encoder():                                  # coop.py:56-67, statement for statement
            x = prompts + model.positional_embedding.type(model.dtype)
            x = x.permute(1, 0, 2)
            x = model.transformer(x)
            x = x.permute(1, 0, 2)
            x = model.ln_final(x).type(model.dtype)
            return x[torch.arange(x.shape[0]), ids.argmax(dim=-1)] @ model.text_projection

        res = {}
        for name, hint in (("every row", None), ("live_rows hint", ids)):
            model.transformer.live_rows = hint
            for _ in range(3):
                out = text_encoder()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                out = text_encoder()
            torch.cuda.synchronize()
            res[name] = ((time.perf_counter() - t0) / 20, out.float())
        gap = float((res["every row"][1] - res["live_rows hint"][1]).abs().max()) / float(res["every row"][1].abs().max())
        print(f"{G} C={Cn}: every row {res['every row'][0]*1e3:.2f} ms, with the hint ({model.live_rows(ids)} of {model.context_length} rows) "
              f"{res['live_rows hint'][0]*1e3:.2f} ms; max |d feature| / max |feature| {gap:.1e}", flush=True)
