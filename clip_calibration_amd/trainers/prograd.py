"""ProGrad (reference trainers/classification/prograd.py:230-289) -- inference forward only.

At test time ProGrad IS CoOp: ``CustomCLIP.forward`` (prograd.py:272-289) runs the prompt learner's ``[SOS | ctx | class]``
splice through the text tower and returns the same 3-tuple; the gradient projection that distinguishes the method lives
in the training loop (prograd.py:291-306, 411-450).  The frozen zero-shot teacher ``CLIP`` (prograd.py:230-259) is
``ZeroshotCLIP`` with the hand-written templates."""
from __future__ import annotations

from .coop import CustomCLIP  # noqa: F401  (same forward, same cache)
from .zsclip import ZeroshotCLIP as CLIP  # noqa: F401
