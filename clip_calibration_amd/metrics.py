"""Expected Calibration Error, host side (reference tools/metrics.py:90-130).

The per-sample work (softmax top-1, binning, per-bin sums) runs on the GPU (``clipmi_logits`` +
``clipmi_ece_accumulate``); this module only turns the 3*(n_bins+1) accumulated numbers into the scalar, reproducing
the reference's digitize/histogram edge quirk.  ``ECE`` keeps the reference's name and signature for callers that
already hold (conf, pred, gt) arrays on the host.
"""
from __future__ import annotations

import numpy as np


def digitize_bins(conf: np.ndarray, n_bins: int) -> np.ndarray:
    """np.digitize(conf, linspace(0,1,n_bins+1)) - 1 (tools/metrics.py:104-105): bin n_bins holds conf == 1.0."""
    edges = np.linspace(0, 1, n_bins + 1)
    return np.searchsorted(edges, np.asarray(conf), side="right") - 1


def bin_statistics(conf, pred, gt, n_bins: int = 10) -> np.ndarray:
    """float64 [3, n_bins+1] = (count, sum_conf, sum_correct) per digitize bin -- what the device kernel accumulates."""
    conf = np.asarray(conf, dtype=np.float64)
    which = digitize_bins(conf, n_bins)
    out = np.zeros((3, n_bins + 1))
    np.add.at(out[0], which, 1.0)
    np.add.at(out[1], which, conf)
    np.add.at(out[2], which, (np.asarray(pred) == np.asarray(gt)).astype(np.float64))
    return out


def ece_from_bins(bins: np.ndarray, n_bins: int = 10) -> float:
    """Scalar ECE from accumulated (count, sum_conf, sum_correct).  Per-bin means come from the digitize bins
    0..n_bins-1; the weights follow np.histogram, whose closed last edge puts conf == 1.0 into bin n_bins-1
    (tools/metrics.py:108-128)."""
    b = np.asarray(bins, dtype=np.float64).reshape(3, n_bins + 1)
    count, s_conf, s_corr = b[0, :n_bins], b[1, :n_bins], b[2, :n_bins]
    total = b[0].sum()
    if total == 0:
        return float("nan")
    nz = count > 0
    acc = np.where(nz, s_corr / np.where(nz, count, 1), 0.0)
    avg = np.where(nz, s_conf / np.where(nz, count, 1), 0.0)
    weights = count.copy()
    weights[n_bins - 1] += b[0, n_bins]
    return float(np.sum(weights / total * np.abs(avg - acc)))


def ECE(conf, pred, gt, conf_bin_num: int = 10) -> float:
    return ece_from_bins(bin_statistics(conf, pred, gt, conf_bin_num), conf_bin_num)


def mce_from_bins(bins: np.ndarray, n_bins: int = 10) -> float:
    """The reference's "MCE" (tools/metrics.py:181-208): bins from digitize(conf, linspace(0,1,n+1)[1:-1]) -- so conf == 1.0
    belongs to the last bin and DOES enter its means -- and the per-bin gap weighted by the bin's share:
    max_b |mean acc_b - mean conf_b| * count_b / N  =  max_b |sum_correct_b - sum_conf_b| / N."""
    b = np.asarray(bins, dtype=np.float64).reshape(3, n_bins + 1).copy()
    b[:, n_bins - 1] += b[:, n_bins]
    b = b[:, :n_bins]
    total = b[0].sum()
    if total == 0:
        return float("nan")
    nz = b[0] > 0
    return float(np.max(np.abs(b[2][nz] - b[1][nz])) / total)


def MCE(conf, pred, gt, conf_bin_num: int = 10) -> float:
    return mce_from_bins(bin_statistics(conf, pred, gt, conf_bin_num), conf_bin_num)
