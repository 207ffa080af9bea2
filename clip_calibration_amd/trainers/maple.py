"""MaPLe (reference trainers/classification/maple.py:51-216) -- inference forward only."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..model import CLIP
from .coop import CustomCLIP as _CoOpCLIP, TextEncoder  # noqa: F401


class MultiModalPromptLearner(nn.Module):
    """maple.py:77-187: shallow text ctx [n_ctx,Dt], its Linear(Dt->Dv) image-side projection, and
    (PROMPT_DEPTH-1) deep text prompts each with its own Linear(Dt->Dv).  The few [n_ctx, D] x [D, Dv] projections are
    tiny host-controlled ops (2 x 512 x 768) kept in torch, as in the reference; ``forward`` returns the same 4-tuple
    (prompts, shared_ctx, deep text prompts, deep visual prompts)."""

    def __init__(self, clip_model: CLIP, tokenized_prompts: torch.Tensor, n_ctx: int = 2, prompt_depth: int = 9, seed: int = 0):
        super().__init__()
        assert prompt_depth >= 1, "For MaPLe, PROMPT_DEPTH should be >= 1"
        dev, dtype = clip_model.device, clip_model.dtype
        dt = clip_model.ln_final.weight.shape[0]
        dv = clip_model.visual.conv1.weight.shape[0]
        g = torch.Generator().manual_seed(seed)
        self.ctx = nn.Parameter((0.02 * torch.randn(n_ctx, dt, generator=g)).to(dev, dtype))
        self.proj = nn.Linear(dt, dv).to(dev).half()
        self.compound_prompts_text = nn.ParameterList(
            [nn.Parameter((0.02 * torch.randn(n_ctx, dt, generator=g)).to(dev)) for _ in range(prompt_depth - 1)])
        self.compound_prompt_projections = nn.ModuleList([nn.Linear(dt, dv).to(dev) for _ in range(prompt_depth - 1)])
        tokenized_prompts = tokenized_prompts.to(dev)
        with torch.no_grad():
            embedding = clip_model.token_embedding(tokenized_prompts).type(dtype)
        self.register_buffer("token_prefix", embedding[:, :1, :])
        self.register_buffer("token_suffix", embedding[:, 1 + n_ctx:, :])
        self.n_cls, self.n_ctx = tokenized_prompts.shape[0], n_ctx
        self.tokenized_prompts = tokenized_prompts

    def forward(self):
        ctx = self.ctx.unsqueeze(0).expand(self.n_cls, -1, -1)
        prompts = torch.cat([self.token_prefix, ctx.to(self.token_prefix.dtype), self.token_suffix], dim=1)
        visual_deep = [layer(p) for layer, p in zip(self.compound_prompt_projections, self.compound_prompts_text)]
        return prompts, self.proj(self.ctx.to(self.proj.weight.dtype)), list(self.compound_prompts_text), visual_deep


class CustomCLIP(_CoOpCLIP):
    """maple.py:190-216: text tower with deep text prompts, image tower with shared_ctx + deep visual prompts."""

    def __init__(self, clip_model: CLIP, tokenized_prompts: torch.Tensor, n_ctx: int = 2, prompt_depth: int = 9,
                 logit_scale=None, cache_text_features: bool = True, seed: int = 0):
        nn.Module.__init__(self)
        self.prompt_learner = MultiModalPromptLearner(clip_model, tokenized_prompts, n_ctx, prompt_depth, seed)
        self.tokenized_prompts = self.prompt_learner.tokenized_prompts
        self.image_encoder = clip_model.visual
        self.text_encoder = TextEncoder(clip_model)
        self.logit_scale = clip_model.logit_scale
        self.dtype = clip_model.dtype
        object.__setattr__(self, "clip_model", clip_model)
        self._fixed_scale = logit_scale
        self.cache_text_features = cache_text_features
        self._cache_key = None
        self._cache = None

    def _text_inputs(self):
        prompts, _, deep_t, _ = self.prompt_learner()
        return prompts, deep_t, self.prompt_learner.n_ctx

    def _image_features(self, image: torch.Tensor) -> torch.Tensor:
        _, shared_ctx, _, deep_v = self.prompt_learner()
        return self.clip_model.image_features_f32(image, shared_ctx, deep_v)
