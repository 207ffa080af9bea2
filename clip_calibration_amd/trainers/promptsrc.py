"""PromptSRC / IVLP (reference trainers/classification/promptsrc.py:73-214) -- inference forward only.

At test time PromptSRC is CoOp's prompt splice (``[SOS | ctx | class tokens]``, promptsrc.py:141-171) run through a CLIP
built with the IVLP design, whose blocks carry their own per-layer prompt tokens on both towers (clip/model.py:191-256).
Those tokens live in the model (``build_model(..., {"trainer": "IVLP", ...})``), so the forward is CoOp's; the only
addition is that the text-feature cache also watches the model's text-side tokens.  The frozen teacher branch
(``fixed_embeddings``, ``ZS_image_encoder``) is read by the training loss only (promptsrc.py:197-210)."""
from __future__ import annotations

import torch

from .coop import CustomCLIP as _CoOpCLIP


class CustomCLIP(_CoOpCLIP):
    def __init__(self, clip_model, tokenized_prompts, n_ctx: int = 4, **kw):
        if clip_model.design_details.get("trainer") != "IVLP" or int(clip_model.design_details.get("language_depth", 0)) < 1:
            raise ValueError("PromptSRC needs a CLIP built with design_details trainer='IVLP' and language_depth >= 1 "
                             "(promptsrc.py:77-80)")
        super().__init__(clip_model, tokenized_prompts, n_ctx=n_ctx, **kw)

    def _cache_params(self):
        deep, _ = self.clip_model.ivlp_text_prompts()
        return list(self.prompt_learner.parameters()) + list(deep or [])
