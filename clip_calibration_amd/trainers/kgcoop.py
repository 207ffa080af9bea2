"""KgCoOp (reference trainers/classification/kgcoop.py:90-271) -- inference forward only.

At test time KgCoOp is CoOp with ``ctx`` initialised from the embedding of "a photo of a" (n_ctx = 4,
kgcoop.py:102-112) plus a stored zero-shot text embedding ``ori_embedding`` (kgcoop.py:151-165) that only the training
loss reads.  ``forward`` (kgcoop.py:246-259) returns the same 3-tuple as CoOp."""
from __future__ import annotations

import torch

from .. import ops
from .coop import CustomCLIP as _CoOpCLIP


class CustomCLIP(_CoOpCLIP):
    def __init__(self, clip_model, tokenized_prompts, zeroshot_tokenized_prompts=None, ctx_init_ids=None, n_ctx: int = 4,
                 w: float = 8.0, **kw):
        super().__init__(clip_model, tokenized_prompts, n_ctx=n_ctx, ctx_init_ids=ctx_init_ids, **kw)
        self.w = w
        self.ori_embedding = None
        if zeroshot_tokenized_prompts is not None:
            with torch.no_grad():  # kgcoop.py:160-165: encode_text of the hand-written prompts, L2-normalised
                self.ori_embedding = ops.l2_normalize(clip_model.text_features_f32(zeroshot_tokenized_prompts.to(clip_model.device)))
