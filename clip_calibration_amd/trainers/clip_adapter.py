"""CLIP-Adapter (reference trainers/classification/clip_adapter.py:138-187) -- inference forward only.

A two-layer bias-free bottleneck (E -> E/4 -> E, ReLU after each) on the un-normalised image features, blended with them
by ``ratio`` (clip_adapter.py:170-172); the text side is CoOp's prompt splice through the text tower (clip_adapter.py:174-176;
with the shipped config the context is the frozen hand-written template)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .coop import CustomCLIP as _CoOpCLIP


class Adapter(nn.Module):
    def __init__(self, c_in: int, reduction: int = 4):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(c_in, c_in // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(c_in // reduction, c_in, bias=False), nn.ReLU(inplace=True))


class CustomCLIP(_CoOpCLIP):
    def __init__(self, clip_model, tokenized_prompts, n_ctx: int = 16, ratio: float = 0.2, **kw):
        super().__init__(clip_model, tokenized_prompts, n_ctx=n_ctx, **kw)
        self.adapter = Adapter(clip_model.visual.output_dim, 4).to(clip_model.device, clip_model.dtype)
        self.ratio = ratio

    def _image_features(self, image: torch.Tensor) -> torch.Tensor:
        f = self.clip_model.image_features_f32(image)
        return ops.adapter_blend(f, self.adapter.fc[0].weight.float(), self.adapter.fc[2].weight.float(), self.ratio)
