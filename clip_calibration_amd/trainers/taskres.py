"""TaskRes (reference trainers/classification/taskres.py:96-210) -- inference forward only.

The classifier is ``base_text_features + alpha * text_feature_residuals`` (taskres.py:105-106), where the base features are
the text-encoder outputs of the hand-written templates averaged per class (taskres.py:109-135, NOT normalised before the
mean) and the residual is the only learned tensor.  ``forward`` normalises both sides and multiplies (taskres.py:191-210;
the reference returns the logits alone, the mirror the trainer-level 3-tuple)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..model import CLIP


class TaskResLearner(nn.Module):
    def __init__(self, base_text_features: torch.Tensor, alpha: float = 0.5):
        super().__init__()
        self.alpha = alpha
        self.register_buffer("base_text_features", base_text_features)
        self.text_feature_residuals = nn.Parameter(torch.zeros_like(base_text_features))

    def forward(self) -> torch.Tensor:
        return ops.scale_add(self.base_text_features, self.text_feature_residuals, self.alpha)


class CustomCLIP(nn.Module):
    def __init__(self, clip_model: CLIP, tokenized_templates: torch.Tensor, alpha: float = 0.5, logit_scale: Optional[float] = None):
        """``tokenized_templates``: [C, T, 77] token ids of T hand-written templates per class (T = 1 for every dataset but
        ImageNet); the base feature of a class is the plain mean of its T text-encoder outputs."""
        super().__init__()
        if tokenized_templates.dim() == 2:
            tokenized_templates = tokenized_templates.unsqueeze(1)
        C, T, L = tokenized_templates.shape
        with torch.no_grad():
            feats = clip_model.text_features_f32(tokenized_templates.reshape(C * T, L).to(clip_model.device))
            base = ops.group_mean(feats, T)
        self.prompt_learner = TaskResLearner(base, alpha)
        self.logit_scale = clip_model.logit_scale
        self.dtype = clip_model.dtype
        object.__setattr__(self, "clip_model", clip_model)
        self._fixed_scale = logit_scale

    @property
    def scale(self) -> float:
        return float(self._fixed_scale) if self._fixed_scale is not None else float(self.logit_scale.detach().exp())

    @torch.no_grad()
    def forward(self, image: torch.Tensor, label=None, dac_conf: Optional[torch.Tensor] = None, want_conf_pred: bool = False):
        text_features = ops.l2_normalize(self.prompt_learner())
        logits, image_features, conf, pred = ops.fused_tail(self.clip_model.image_features_f32(image), text_features, self.scale,
                                                            dac_conf, want_conf_pred)
        if want_conf_pred:
            return logits, image_features, text_features, conf, pred
        return logits, image_features, text_features
