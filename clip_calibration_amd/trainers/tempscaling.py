"""TempScaling (reference trainers/calibration/tempscaling.py:31-59) -- the forward; the 1-parameter SGD fit stays in
torch autograd on the caller's side."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class ScaleLearner(nn.Module):
    """tempscaling.py:31-41: one scalar ``logit_scale`` initialised to 4.6052 (= ln 100)."""

    def __init__(self, dtype=torch.float32, init: float = 4.6052, device=None):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.tensor(init, dtype=dtype, device=device))

    def forward(self):
        return self.logit_scale.exp()


class CustomCLIPCalibration(nn.Module):
    """tempscaling.py:44-59: ``logits_encoder`` is a base model that emits cosine logits (logit_scale = 1.0); the learnt
    scale multiplies the normalised image features before the text matmul."""

    def __init__(self, base_model, init: float = 4.6052):
        super().__init__()
        self.logits_encoder = base_model
        self.dtype = getattr(base_model, "dtype", torch.float32)
        self.scale_learner = ScaleLearner(torch.float32, init)

    @torch.no_grad()
    def forward(self, image, label=None, dac_conf=None, want_conf_pred: bool = False):
        _, image_features, text_features = self.logits_encoder(image)[:3]
        scale = float(self.scale_learner().detach())
        logits, conf, pred = ops.logits_fused(image_features, text_features, scale, dac_conf, want_conf_pred)
        if want_conf_pred:
            return logits, image_features, text_features, conf, pred
        return logits, image_features, text_features
