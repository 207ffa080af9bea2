"""Device-side stand-in for VLClassification.process/evaluate and VLCalibration.predict on the ECE branch
(reference evaluators/vl_evaluator.py:40-51,59-102; trainers/calibration/vl_calibrator.py:83-109) -- SURVEY f-1.

The reference copies logits, labels and both feature matrices to python lists every batch (3 D2H syncs + ``tolist()``);
here (conf, pred) come out of the logits kernel and only 3*(n_bins+1) float64 accumulators live on the device until
``evaluate``.  With ``keep_samples=True`` the per-sample (conf, pred, label) vectors -- 16 B per sample -- are kept on
the device as well and copied to the host ONCE in ``evaluate`` for the metrics that need the samples themselves
(macro-F1, AdaptiveECE's equal-mass bins, PIECE's proximity bins)."""
from __future__ import annotations

from collections import OrderedDict
from typing import List, Optional

import numpy as np
import torch

from . import ops
from .metrics import AdaptiveECE, PIECE, ece_from_bins, macro_f1, mce_from_bins


class DeviceCalibrationEvaluator:
    def __init__(self, n_bins: int = 10, device="cuda", keep_samples: bool = False, piece_bins: int = 10):
        self.n_bins = n_bins
        self.piece_bins = piece_bins
        self.keep_samples = keep_samples
        self.bins = torch.zeros(3 * (n_bins + 1), dtype=torch.float64, device=device)
        self._conf: List[torch.Tensor] = []
        self._pred: List[torch.Tensor] = []
        self._gt: List[torch.Tensor] = []

    def reset(self):
        self.bins.zero_()
        self._conf, self._pred, self._gt = [], [], []

    def process(self, conf: torch.Tensor, pred: torch.Tensor, gt: torch.Tensor):
        gt = gt.to(conf.device, torch.int64)
        ops.ece_accumulate(conf, pred, gt, self.bins, self.n_bins)
        if self.keep_samples:
            self._conf.append(conf)
            self._pred.append(pred)
            self._gt.append(gt)

    def note_processed(self, conf: torch.Tensor, pred: torch.Tensor, gt: torch.Tensor):
        """The fused tail (ops.fused_tail with ``bins=self.bins``) has already added this batch to the bin accumulators;
        only the optional per-sample vectors are kept here."""
        if self.keep_samples:
            self._conf.append(conf)
            self._pred.append(pred)
            self._gt.append(gt.to(conf.device, torch.int64))

    def merge_from(self, other_bins: torch.Tensor):
        self.bins += other_bins.to(self.bins.device)

    def samples(self):
        """(conf f32, pred i64, gt i64) numpy vectors of everything processed so far: one D2H copy each."""
        if not self.keep_samples:
            raise RuntimeError("evaluator was built with keep_samples=False")
        if not self._conf:
            return np.zeros(0, np.float32), np.zeros(0, np.int64), np.zeros(0, np.int64)
        return (torch.cat(self._conf).cpu().numpy(), torch.cat(self._pred).cpu().numpy().astype(np.int64),
                torch.cat(self._gt).cpu().numpy())

    def evaluate(self, proximity: Optional[np.ndarray] = None) -> "OrderedDict[str, float]":
        """Keys and scaling follow vl_evaluator.py:59-102 (percentages except ``confidence``).  ``macro_f1`` and ``ace``
        need keep_samples; ``piece`` additionally needs the per-sample proximity (base_learner.py:136-137)."""
        b = self.bins.cpu().numpy().reshape(3, self.n_bins + 1)
        total = b[0].sum()
        res = OrderedDict()
        res["accuracy"] = 100.0 * b[2].sum() / total
        res["error_rate"] = 100.0 - res["accuracy"]
        if self.keep_samples:
            conf, pred, gt = self.samples()
            res["macro_f1"] = 100.0 * macro_f1(pred, gt)
        res["confidence"] = b[1].sum() / total
        res["ece"] = 100.0 * ece_from_bins(b, self.n_bins)
        res["mce"] = 100.0 * mce_from_bins(b, self.n_bins)
        if self.keep_samples:
            res["ace"] = 100.0 * AdaptiveECE(conf, pred, gt, self.n_bins)
            if proximity is not None:
                proximity = np.asarray(proximity)
                if proximity.shape[0] != conf.shape[0]:
                    raise ValueError(f"proximity has {proximity.shape[0]} rows for {conf.shape[0]} samples")
                res["piece"] = 100.0 * PIECE(conf, proximity, pred, gt, self.piece_bins, self.n_bins)
        res["total"] = int(total)
        return res
