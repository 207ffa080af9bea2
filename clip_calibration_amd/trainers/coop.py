"""CoOp (reference trainers/classification/coop.py:47-222) -- inference forward only."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..model import CLIP


class TextEncoder(nn.Module):
    """coop.py:47-67.  ``forward(prompts, tokenized_prompts)``: the reference adds the positional embedding, runs
    ``clip_model.transformer``, ``ln_final``, gathers the EOT row and multiplies by ``text_projection``; here the same
    chain is one fused device call (clipmi_text_encoder)."""

    def __init__(self, clip_model: CLIP):
        super().__init__()
        object.__setattr__(self, "clip_model", clip_model)
        self.dtype = clip_model.dtype

    def forward(self, prompts: torch.Tensor, tokenized_prompts: torch.Tensor, compound_prompts_deeper_text=None,
                n_ctx: int = 0, flags: int = 0, seq_rows: Optional[int] = None) -> torch.Tensor:
        return self.clip_model.text_encoder_f32(prompts, tokenized_prompts, compound_prompts_deeper_text, n_ctx, flags=flags, seq_rows=seq_rows)


def n_ctx_from_init_ids(ctx_init_ids: torch.Tensor, context_length: int) -> int:
    """Number of context words of a CTX_INIT prompt from its token ids ``[1, L]`` = ``[SOT, w1..wn, EOT, 0...]``: the
    reference counts ``len(ctx_init.split(" "))`` (coop.py:82-84); here it is the EOT position (the largest id, as in
    ``clip.tokenize``) minus one, so both the zero-padded ``[1, 77]`` tensor ``tokenize`` returns and an unpadded
    ``[1, n+2]`` slice give n.  (For the templates the configs ship -- "a photo of a" -- every word is one token.)"""
    ids = ctx_init_ids.reshape(-1, ctx_init_ids.shape[-1])[0]
    n_ctx = int(ids.argmax()) - 1
    if not (1 <= n_ctx and 1 + n_ctx < context_length):
        raise ValueError(f"CTX_INIT ids give n_ctx={n_ctx}: expected [SOT, words.., EOT, padding] with 1 <= n_ctx < {context_length - 1}")
    return n_ctx


class PromptLearner(nn.Module):
    """coop.py:70-144 with CLASS_TOKEN_POSITION == 'end' (the shipped configs): prompts = [SOS | ctx | class tokens, EOS, pad].
    ``tokenized_prompts`` are the ids of ``"X X ... X {classname}."`` (coop.py:107-112)."""

    def __init__(self, clip_model: CLIP, tokenized_prompts: torch.Tensor, n_ctx: int = 16, csc: bool = False,
                 ctx_init_ids: Optional[torch.Tensor] = None, seed: int = 0):
        super().__init__()
        dtype = clip_model.dtype
        dev = clip_model.device
        ctx_dim = clip_model.ln_final.weight.shape[0]
        n_cls = tokenized_prompts.shape[0]
        tokenized_prompts = tokenized_prompts.to(dev)
        with torch.no_grad():
            if ctx_init_ids is not None:  # CTX_INIT: embedding of the given words (coop.py:82-90)
                emb = clip_model.token_embedding(ctx_init_ids.to(dev)).type(dtype)
                n_ctx = n_ctx_from_init_ids(ctx_init_ids, tokenized_prompts.shape[-1])
                ctx_vectors = emb[0, 1:1 + n_ctx, :].clone()
            else:
                g = torch.Generator().manual_seed(seed)
                shape = (n_cls, n_ctx, ctx_dim) if csc else (n_ctx, ctx_dim)
                ctx_vectors = (0.02 * torch.randn(*shape, generator=g)).to(dev, dtype)
            embedding = clip_model.token_embedding(tokenized_prompts).type(dtype)
        if not 1 + n_ctx < tokenized_prompts.shape[-1]:
            raise ValueError(f"n_ctx={n_ctx} does not fit a context of {tokenized_prompts.shape[-1]} tokens")
        self.ctx = nn.Parameter(ctx_vectors)
        self.register_buffer("token_prefix", embedding[:, :1, :])            # SOS
        self.register_buffer("token_suffix", embedding[:, 1 + n_ctx:, :])    # class tokens, EOS, padding
        self.n_cls, self.n_ctx = n_cls, n_ctx
        self.tokenized_prompts = tokenized_prompts
        self.class_token_position = "end"

    def forward(self) -> torch.Tensor:
        ctx = self.ctx
        if ctx.dim() == 2:
            ctx = ctx.unsqueeze(0).expand(self.n_cls, -1, -1)
        return torch.cat([self.token_prefix, ctx.to(self.token_prefix.dtype), self.token_suffix], dim=1)  # layout only


class CustomCLIP(nn.Module):
    """coop.py:192-222.  eval ``forward(image) -> (logits, image_features, text_features)``.

    The reference re-runs the 12-layer text tower on every batch although ``ctx`` is frozen at test time; the text
    features are cached here and recomputed only when ``ctx`` changes (same outputs; precedent in the reference:
    ProDA's set_classifier, proda.py:315-333).  ``logit_scale=1.0`` gives the cosine-logit base model of
    trainers/calibration/base_model/coop.py:222-224.

    ``cache_text_features=False`` is the reference's own schedule (coop.py:208-210: prompt learner + text tower on every
    batch).  The text tower is then per-batch work, and ``text_stream_f16`` (None = "when not cached") runs it on the fp16
    residual stream -- the reference's own GPU precision, clip/model.py:186-187 -- through the per-call flag
    CLIPMI_CALL_STREAM_F16, as the CoCoOp mirror does: -17 % per text tower call at 500 classes
    (profiles/r03_bench_coop_dac.json).  Cached features keep the fp32 stream: they are computed once and feed every logit."""

    text_stream_f16: Optional[bool] = None   # class default for mirrors that build themselves (MaPLe)

    def __init__(self, clip_model: CLIP, tokenized_prompts: torch.Tensor, n_ctx: int = 16, csc: bool = False,
                 logit_scale: Optional[float] = None, cache_text_features: bool = True,
                 text_stream_f16: Optional[bool] = None, **kw):
        super().__init__()
        self.text_stream_f16 = text_stream_f16
        self.prompt_learner = PromptLearner(clip_model, tokenized_prompts, n_ctx, csc, **kw)
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

    @property
    def scale(self) -> float:
        return float(self._fixed_scale) if self._fixed_scale is not None else float(self.logit_scale.detach().exp())

    def _text_inputs(self):
        return self.prompt_learner(), None, 0

    def _text_flags(self) -> int:
        from .. import _lib
        want = (not self.cache_text_features) if self.text_stream_f16 is None else self.text_stream_f16
        m = self.clip_model
        # a model (or process) set to residual_f16 != 2 or ln_fold = 0 has chosen its streams explicitly: no per-call override
        if want and m.get_option("residual_f16") == 2 and m.get_option("ln_fold") == 1:
            return _lib.CALL_STREAM_F16
        return _lib.CALL_DEFAULT

    def _cache_params(self):
        """Everything the cached text features depend on (besides the frozen tower weights)."""
        return list(self.prompt_learner.parameters())

    @torch.no_grad()
    def text_features(self) -> torch.Tensor:
        key = tuple((p.data_ptr(), p._version) for p in self._cache_params())
        if self.cache_text_features and key == self._cache_key and self._cache is not None:
            return self._cache
        prompts, deep, n_ctx = self._text_inputs()
        # the live token rows of the class list, from the OWNER's copy of the ids and the owner's table (one read-back per prompt set, ever): under
        # nn.DataParallel a replica's buffers are re-broadcast -- new tensors -- on every forward, and a per-forward bound would sync every device
        m = self.clip_model
        rows = m._home().live_rows(self.prompt_learner.tokenized_prompts, n_ctx if deep else 0) if hasattr(m, "_home") else None
        tf = ops.l2_normalize(self.text_encoder(prompts, self.tokenized_prompts, deep, n_ctx, flags=self._text_flags(), seq_rows=rows))
        self._cache_key, self._cache = key, tf
        return tf

    def _image_features(self, image: torch.Tensor) -> torch.Tensor:
        return self.clip_model.image_features_f32(image)

    overlap_towers: bool = True   # per-batch text tower (cache_text_features=False): run it on a side stream beside the image tower

    @torch.no_grad()
    def towers(self, image: torch.Tensor):
        """(image features fp32 un-normalised, text features fp32 normalised) of one batch.  With the reference's per-batch schedule
        (coop.py:208-210: prompt learner + text tower on every batch) the two towers are independent until the logits, so the text tower
        is issued on a side stream and the image tower on the caller's; the caller's stream then waits for the side stream.  They use
        separate workspaces (the model's "text" and "vision" buffers)."""
        if self.cache_text_features or not self.overlap_towers or not image.is_cuda:
            text_features = self.text_features()
            return self._image_features(image), text_features
        # Weight binding is lazy and packs BOTH towers with torch ops on whatever stream is current: do it here, on the caller's stream,
        # so that the side stream (which waits for this point) and the caller's stream both see finished operands.
        self.clip_model._resident(image.device)._ensure_bound()
        cur = torch.cuda.current_stream(image.device)
        side = getattr(self, "_side_stream", None)
        if side is None or side.device != image.device:
            side = torch.cuda.Stream(device=image.device)
            object.__setattr__(self, "_side_stream", side)
        side.wait_stream(cur)                       # the prompt learner's parameters may have been written on the caller's stream
        with torch.cuda.stream(side):
            text_features = self.text_features()
        image_features = self._image_features(image)
        cur.wait_stream(side)
        text_features.record_stream(cur)
        return image_features, text_features

    @torch.no_grad()
    def forward(self, image: torch.Tensor, label=None, dac_conf: Optional[torch.Tensor] = None, want_conf_pred: bool = False):
        image_features_raw, text_features = self.towers(image)
        logits, image_features, conf, pred = ops.fused_tail(image_features_raw, text_features, self.scale, dac_conf,
                                                            want_conf_pred)
        if want_conf_pred:
            return logits, image_features, text_features, conf, pred
        return logits, image_features, text_features
