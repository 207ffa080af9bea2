"""CoCoOp (reference trainers/classification/cocoop.py:71-199) -- inference forward only.

Every image gets its own context: ``ctx + meta_net(image_features)`` (cocoop.py:154-161), hence its own C prompts and its
own pass through the text tower -- B*C prompts per batch, which makes the TEXT tower the dominant cost (5.96 GFLOP per
prompt against 35 GFLOP per image).  The reference loops over the images in Python and runs the tower on C prompts at
a time (cocoop.py:193-198); here the prompts of several images form one text-encoder call (``prompts_per_call``), the
meta-net, the prompt splice and the per-image normalise + dot product are HIP kernels, and nothing returns to the host.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..model import CLIP
from .coop import TextEncoder


class PromptLearner(nn.Module):
    """cocoop.py:71-171: ``ctx`` [n_ctx, D] plus ``meta_net`` = Linear(E, E//16) - ReLU - Linear(E//16, D).  ``forward``
    returns the shifted contexts [B, n_ctx, D]; the [B, C, 77, D] prompt tensor the reference stacks is never built whole."""

    def __init__(self, clip_model: CLIP, tokenized_prompts: torch.Tensor, n_ctx: int = 4,
                 ctx_init_ids: Optional[torch.Tensor] = None, seed: int = 0):
        super().__init__()
        dtype, dev = clip_model.dtype, clip_model.device
        ctx_dim = clip_model.ln_final.weight.shape[0]
        vis_dim = clip_model.visual.output_dim
        tokenized_prompts = tokenized_prompts.to(dev)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            if ctx_init_ids is not None:      # CTX_INIT "a photo of a" (cocoop.py:85-93)
                emb = clip_model.token_embedding(ctx_init_ids.to(dev)).type(dtype)
                n_ctx = n_ctx_from_init_ids(ctx_init_ids, tokenized_prompts.shape[-1])
                ctx_vectors = emb[0, 1:1 + n_ctx, :].clone()
            else:
                ctx_vectors = (0.02 * torch.randn(n_ctx, ctx_dim, generator=g)).to(dev, dtype)
            embedding = clip_model.token_embedding(tokenized_prompts).type(dtype)
        self.ctx = nn.Parameter(ctx_vectors)
        self.meta_net = nn.Sequential(OrderedDict([("linear1", nn.Linear(vis_dim, vis_dim // 16)), ("relu", nn.ReLU(inplace=True)),
                                                   ("linear2", nn.Linear(vis_dim // 16, ctx_dim))])).to(dev, dtype)
        self.register_buffer("token_prefix", embedding[:, :1, :])
        self.register_buffer("token_suffix", embedding[:, 1 + n_ctx:, :])
        self.register_buffer("_base", embedding, persistent=False)     # [C,77,D]: rows 1..n_ctx are overwritten per image
        self.n_cls, self.n_ctx = tokenized_prompts.shape[0], n_ctx
        self.tokenized_prompts = tokenized_prompts

    def base_embedding(self) -> torch.Tensor:
        return self._base

    def forward(self, im_features: torch.Tensor) -> torch.Tensor:
        m = self.meta_net
        return ops.cocoop_ctx(im_features.float(), m.linear1.weight.float(), m.linear1.bias.float(), m.linear2.weight.float(),
                              m.linear2.bias.float(), self.ctx.float())


class CustomCLIP(nn.Module):
    """cocoop.py:174-199.  eval ``forward(image) -> (logits [B,C], image_features [B,E], text_features [C,E])`` where, as in
    the reference, ``text_features`` are those of the LAST image of the batch."""

    def __init__(self, clip_model: CLIP, tokenized_prompts: torch.Tensor, n_ctx: int = 4, logit_scale: Optional[float] = None,
                 prompts_per_call: int = 4096, text_stream_f16: bool = True, **kw):
        super().__init__()
        # Here the TEXT tower is the hot path (B * C prompts per batch, cocoop.py:186-197).  text_stream_f16: run it on the fp16
        # residual stream like the image tower (the reference's own GPU precision: clip/model.py:186-187 adds in fp16) -- +20 %
        # prompts/s (profiles/r02_text_tower_f16_stream.txt).  The library default keeps the text tower's stream in fp32 because
        # zero-shot / CoOp text features are computed once and feed every logit.  It is a PER-CALL flag of the tower call
        # (CLIPMI_CALL_STREAM_F16): no library state is touched, the same model handle keeps serving fp32-stream text features;
        # a model (or process) set to residual_f16 = 0 or ln_fold = 0 keeps the fp32 stream here too.
        self.text_stream_f16 = text_stream_f16
        self.prompt_learner = PromptLearner(clip_model, tokenized_prompts, n_ctx, **kw)
        self.tokenized_prompts = self.prompt_learner.tokenized_prompts
        self.image_encoder = clip_model.visual
        self.text_encoder = TextEncoder(clip_model)
        self.logit_scale = clip_model.logit_scale
        self.dtype = clip_model.dtype
        object.__setattr__(self, "clip_model", clip_model)
        self._fixed_scale = logit_scale
        self.prompts_per_call = prompts_per_call

    @property
    def scale(self) -> float:
        return float(self._fixed_scale) if self._fixed_scale is not None else float(self.logit_scale.detach().exp())

    @torch.no_grad()
    def per_image_text_features(self, image_features: torch.Tensor) -> torch.Tensor:
        """Un-normalised text-encoder outputs [B, C, E] for the given L2-normalised image features."""
        pl = self.prompt_learner
        B, Cn = image_features.shape[0], pl.n_cls
        ctx_shifted = pl(image_features)
        out = torch.empty(B, Cn, self.clip_model.geometry.embed_dim, dtype=torch.float32, device=image_features.device)
        step = max(1, self.prompts_per_call // Cn)
        from .. import _lib
        m = self.clip_model
        f16 = self.text_stream_f16 and m.get_option("residual_f16") == 2 and m.get_option("ln_fold") == 1   # 2 = image tower only (the default)
        flags = _lib.CALL_STREAM_F16 if f16 else _lib.CALL_DEFAULT
        rows = m.live_rows(self.tokenized_prompts)      # of the class list (kept with it: no per-batch read-back); every image repeats it
        for lo in range(0, B, step):
            nb = min(step, B - lo)
            prompts = ops.cocoop_prompts(pl.base_embedding(), ctx_shifted[lo:lo + nb])
            ids = self.tokenized_prompts.repeat(nb, 1)                       # EOT index plumbing
            out[lo:lo + nb] = self.text_encoder(prompts, ids, flags=flags, seq_rows=rows).view(nb, Cn, -1)
        return out

    @torch.no_grad()
    def forward(self, image: torch.Tensor, label=None, dac_conf: Optional[torch.Tensor] = None, want_conf_pred: bool = False):
        image_features = ops.l2_normalize(self.clip_model.image_features_f32(image))
        txt = self.per_image_text_features(image_features)
        logits, conf, pred, text_features = ops.logits_per_image(image_features, txt, self.scale, dac_conf, want_conf_pred)
        if want_conf_pred:
            return logits, image_features, text_features, conf, pred
        return logits, image_features, text_features
