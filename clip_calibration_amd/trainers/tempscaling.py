"""TempScaling (reference trainers/calibration/tempscaling.py:31-59) -- the forward, and the hand-off to the 1-parameter
SGD fit, which stays in torch autograd on the caller's side: ``forward_train`` returns
``scale_learner() * cosine_logits`` with the cosine logits from the HIP path as a constant and the scalar as the only
leaf, so ``F.cross_entropy(logits, label).backward()`` is exactly the reference's step (tempscaling.py:146-158)."""
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

    def forward_train(self, image, label=None):
        """tempscaling.py:53-56 under autograd: ``logit_scale * image_features @ text_features.t()``.  The towers and the
        [B, C] cosine matmul run on the HIP path (no graph: every tower weight is frozen there, tempscaling.py:93-106);
        the product with the learnt scalar is the one differentiable operation, so d loss / d logit_scale is the
        reference's.  Returns (logits, image_features, text_features) like ``forward``."""
        with torch.no_grad():
            _, image_features, text_features = self.logits_encoder(image)[:3]
            cosine, _, _ = ops.logits_fused(image_features, text_features, 1.0, None, False)
        return self.scale_learner() * cosine, image_features, text_features

    @torch.no_grad()
    def forward(self, image, label=None, dac_conf=None, want_conf_pred: bool = False):
        _, image_features, text_features = self.logits_encoder(image)[:3]
        scale = float(self.scale_learner().detach())
        logits, conf, pred = ops.logits_fused(image_features, text_features, scale, dac_conf, want_conf_pred)
        if want_conf_pred:
            return logits, image_features, text_features, conf, pred
        return logits, image_features, text_features
