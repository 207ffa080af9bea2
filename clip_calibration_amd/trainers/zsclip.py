"""ZeroshotCLIP (reference trainers/classification/zsclip.py:73-102)."""
from __future__ import annotations

import torch

from .. import ops
from ..model import CLIP


class ZeroshotCLIP:
    """``build_model`` encodes the class prompts once and keeps the L2-normalised text features (zsclip.py:76-95);
    ``model_inference`` = image tower -> normalise -> ``exp(logit_scale) * img @ txt^T`` (zsclip.py:97-102).
    Everything stays fp32 on the device; the 3-tuple is the trainer contract of base_learner.py:86."""

    def __init__(self, clip_model: CLIP, tokenized_prompts: torch.Tensor, logit_scale: float | None = None):
        self.clip_model = clip_model
        self._fixed_scale = logit_scale          # 1.0 for the calibration base model (base_model/zsclip.py)
        self.build_model(tokenized_prompts)

    def build_model(self, tokenized_prompts: torch.Tensor) -> None:
        prompts = tokenized_prompts.to(self.clip_model.device)
        with torch.no_grad():
            self.text_features = ops.l2_normalize(self.clip_model.text_features_f32(prompts))

    @property
    def scale(self) -> float:
        return float(self._fixed_scale) if self._fixed_scale is not None else float(self.clip_model.logit_scale.detach().exp())

    @torch.no_grad()
    def model_inference(self, image: torch.Tensor, dac_conf: torch.Tensor | None = None, want_conf_pred: bool = False,
                        labels: torch.Tensor | None = None, evaluator=None):
        """zsclip.py:97-102 as image tower + ONE tail launch (normalise, logits, DAC, softmax top-1 and -- when a
        DeviceCalibrationEvaluator and labels are passed -- its ECE bin accumulation)."""
        bins, n_bins = (evaluator.bins, evaluator.n_bins) if evaluator is not None and labels is not None else (None, 0)
        logits, image_features, conf, pred = ops.fused_tail(self.clip_model.image_features_f32(image), self.text_features, self.scale,
                                                            dac_conf, want_conf_pred, True, labels if bins is not None else None, bins, n_bins)
        if bins is not None:
            evaluator.note_processed(conf, pred, labels)
        if want_conf_pred:
            return logits, image_features, self.text_features, conf, pred
        return logits, image_features, self.text_features

    __call__ = model_inference
