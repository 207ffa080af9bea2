"""VPT (reference trainers/classification/vpt.py:70-116) -- inference forward only.

Vision-side prompting: the CLIP is built with design_details trainer='VPT' (prompt tokens inside the image tower,
clip/model.py:361-424), the text side is the fixed hand-written prompts "a photo of a {name}." encoded once
(``FixedEmbeddings``, vpt.py:70-92).  The reference's eval ``forward`` returns the logits alone (vpt.py:105-116); the
mirror returns the trainer-level 3-tuple every other trainer returns (base_learner.py:86), logits first."""
from __future__ import annotations

import torch

from .zsclip import ZeroshotCLIP


class CustomCLIP(ZeroshotCLIP):
    def __init__(self, clip_model, tokenized_prompts: torch.Tensor, logit_scale: float | None = None):
        if clip_model.design_details.get("trainer") != "VPT" or int(clip_model.design_details.get("vision_depth", 0)) < 1:
            raise ValueError("VPT needs a CLIP built with design_details trainer='VPT' and vision_depth >= 1 (vpt.py:39)")
        super().__init__(clip_model, tokenized_prompts, logit_scale)

    @property
    def fixed_embeddings(self) -> torch.Tensor:
        return self.text_features
