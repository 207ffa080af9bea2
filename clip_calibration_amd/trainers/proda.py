"""ProDA (reference trainers/classification/proda.py:76-333) -- inference forward only.

A collection of ``n_prompt`` learned contexts; at test time ``set_classifier`` (proda.py:316-333) runs ALL
n_cls * n_prompt prompts through the text tower once, L2-normalises each feature, and keeps the per-class MEAN (not
re-normalised) as the classifier; ``forward`` (proda.py:309-313) is then image tower -> normalise -> scaled matmul.
Context position varies over the collection (proda.py:110-114): the first quarter of the prompts put the class name in
front of the context, the second quarter in the middle, the rest at the end.

Index plumbing (which embedding row goes where) is torch indexing; the tower, normalisation, ensemble mean and logits
are device kernels."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..model import CLIP
from .coop import TextEncoder


class PromptLearner(nn.Module):
    def __init__(self, clip_model: CLIP, tokenized_prompts: torch.Tensor, n_ctx: int = 16, n_prompt: int = 32, seed: int = 0):
        super().__init__()
        dtype, dev = clip_model.dtype, clip_model.device
        ctx_dim = clip_model.ln_final.weight.shape[0]
        tokenized_prompts = tokenized_prompts.to(dev)
        g = torch.Generator().manual_seed(seed)
        self.ctx = nn.Parameter((0.02 * torch.randn(n_prompt, n_ctx, ctx_dim, generator=g)).to(dev, dtype))
        if n_prompt > 1:    # proda.py:110-114
            pos = [0] * (n_prompt // 4) + [1] * (n_prompt // 4) + [2] * (n_prompt // 2)
        else:
            pos = [2]
        self.register_buffer("pos", torch.tensor(pos, device=dev), persistent=False)
        with torch.no_grad():
            embedding = clip_model.token_embedding(tokenized_prompts).type(dtype)
        self.register_buffer("token_prefix", embedding[:, :1, :])
        self.register_buffer("token_suffix", embedding[:, 1 + n_ctx:, :])
        # tokens of "X X .. X name ." are [SOT, X*n_ctx, name.., '.', EOT]: the name length follows from the EOT position
        self.name_lens = (tokenized_prompts.argmax(dim=-1) - n_ctx - 2).tolist()
        self.n_cls, self.n_ctx, self.n_prompt = tokenized_prompts.shape[0], n_ctx, n_prompt
        self.tokenized_prompts = tokenized_prompts

    def forward(self, infer: bool = True):
        """proda.py:146-222 with infer=True: prompts [n_cls * n_prompt, 77, D] ordered class-major, and inside a class
        [end-position prompts | middle | front] -- the order the reference's final ``torch.cat(..., dim=1)`` produces."""
        ctx, pos = self.ctx, self.pos
        P, n_cls, half = ctx.shape[0], self.n_cls, self.n_ctx // 2
        tokenized = self.tokenized_prompts.unsqueeze(1).repeat(1, P, 1).view(n_cls * P, -1)
        pre, suf = self.token_prefix, self.token_suffix
        ctx_end, ctx_mid, ctx_front = ctx[pos == 2], ctx[pos == 1], ctx[pos == 0]

        def rep(t, n):      # [1, len, D] -> [1, n, len, D]
            return t.unsqueeze(1).expand(-1, n, -1, -1)
        end = torch.cat([rep(pre, ctx_end.shape[0]), ctx_end.unsqueeze(0).expand(n_cls, -1, -1, -1), rep(suf, ctx_end.shape[0])], dim=2)
        mids, fronts = [], []
        for i, nl in enumerate(self.name_lens):
            p_i, cls_i, suf_i = pre[i:i + 1], suf[i:i + 1, :nl], suf[i:i + 1, nl:]
            nm, nf = ctx_mid.shape[0], ctx_front.shape[0]
            mids.append(torch.cat([rep(p_i, nm), ctx_mid[:, :half].unsqueeze(0), rep(cls_i, nm), ctx_mid[:, half:].unsqueeze(0),
                                   rep(suf_i, nm)], dim=2))
            fronts.append(torch.cat([rep(p_i, nf), rep(cls_i, nf), ctx_front.unsqueeze(0), rep(suf_i, nf)], dim=2))
        prompts = torch.cat([end, torch.cat(mids, dim=0), torch.cat(fronts, dim=0)], dim=1)
        return prompts.reshape(n_cls * P, -1, ctx.shape[-1]), tokenized


class CustomCLIP(nn.Module):
    def __init__(self, clip_model: CLIP, tokenized_prompts: torch.Tensor, n_ctx: int = 16, n_prompt: int = 32,
                 logit_scale: Optional[float] = None, prompts_per_call: int = 4096, **kw):
        super().__init__()
        self.n_class, self.n_prompt = tokenized_prompts.shape[0], n_prompt
        self.text_encoder = TextEncoder(clip_model)
        self.prompt_learner = PromptLearner(clip_model, tokenized_prompts, n_ctx, n_prompt, **kw)
        self.image_encoder = clip_model.visual
        self.logit_scale = clip_model.logit_scale
        self.dtype = clip_model.dtype
        object.__setattr__(self, "clip_model", clip_model)
        self._fixed_scale = logit_scale
        self.prompts_per_call = prompts_per_call
        self.text_features: Optional[torch.Tensor] = None

    @property
    def scale(self) -> float:
        return float(self._fixed_scale) if self._fixed_scale is not None else float(self.logit_scale.detach().exp())

    @torch.no_grad()
    def set_classifier(self) -> None:
        """proda.py:316-333 (called by VLBaseLearner.test before the loop, base_learner.py:65-67)."""
        prompts, tokenized = self.prompt_learner(infer=True)
        feats = []
        for lo in range(0, prompts.shape[0], self.prompts_per_call):
            feats.append(self.text_encoder(prompts[lo:lo + self.prompts_per_call].contiguous(), tokenized[lo:lo + self.prompts_per_call]))
        tf = ops.l2_normalize(torch.cat(feats))
        self.text_features = ops.group_mean(tf, self.n_prompt)

    @torch.no_grad()
    def forward(self, image: torch.Tensor, label=None, dac_conf: Optional[torch.Tensor] = None, want_conf_pred: bool = False):
        if self.text_features is None:
            self.set_classifier()
        logits, image_features, conf, pred = ops.fused_tail(self.clip_model.image_features_f32(image), self.text_features, self.scale,
                                                            dac_conf, want_conf_pred)
        if want_conf_pred:
            return logits, image_features, self.text_features, conf, pred
        return logits, image_features, self.text_features
