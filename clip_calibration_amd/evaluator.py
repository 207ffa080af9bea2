"""Device-side stand-in for VLClassification.process/evaluate and VLCalibration.predict on the ECE branch
(reference evaluators/vl_evaluator.py:40-51,59-92; trainers/calibration/vl_calibrator.py:83-109) -- SURVEY f-1.

The reference copies logits, labels and features to python lists every batch; here (conf, pred) come out of the
logits kernel and only 3*(n_bins+1) float64 accumulators plus two counters live on the device until ``evaluate``."""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import ops
from .metrics import ece_from_bins, mce_from_bins


class DeviceCalibrationEvaluator:
    def __init__(self, n_bins: int = 10, device="cuda"):
        self.n_bins = n_bins
        self.bins = torch.zeros(3 * (n_bins + 1), dtype=torch.float64, device=device)

    def reset(self):
        self.bins.zero_()

    def process(self, conf: torch.Tensor, pred: torch.Tensor, gt: torch.Tensor):
        ops.ece_accumulate(conf, pred, gt.to(conf.device, torch.int64), self.bins, self.n_bins)

    def merge_from(self, other_bins: torch.Tensor):
        self.bins += other_bins.to(self.bins.device)

    def evaluate(self) -> "OrderedDict[str, float]":
        b = self.bins.cpu().numpy().reshape(3, self.n_bins + 1)
        total = b[0].sum()
        res = OrderedDict()
        res["accuracy"] = 100.0 * b[2].sum() / total
        res["error_rate"] = 100.0 - res["accuracy"]
        res["confidence"] = b[1].sum() / total
        res["ece"] = 100.0 * ece_from_bins(b, self.n_bins)
        res["mce"] = 100.0 * mce_from_bins(b, self.n_bins)
        res["total"] = int(total)
        return res
