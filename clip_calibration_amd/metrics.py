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


# ---- metrics that need the samples themselves (quantile bins, per-class counts): evaluated once per test() on the host
# from the (conf, pred, gt[, proximity]) vectors the device evaluator kept -- 12-16 B per sample ------------------------

def quantile_bin_index(x, n_bins: int) -> np.ndarray:
    """Ordinal equal-frequency bin of every value: edges = percentiles 0, 100/n, ..., 100 (linear interpolation), edges
    closer than 1e-8 to their left neighbour dropped, index = number of inner edges <= x.  This is what the reference
    obtains from sklearn's KBinsDiscretizer(encode='ordinal', strategy='quantile') at tools/metrics.py:152,228; a
    constant input collapses to a single bin."""
    col = np.asarray(x)
    if col.dtype not in (np.float32, np.float64):
        col = col.astype(np.float64)
    if col.size == 0:
        return np.zeros(0, dtype=np.int64)
    if col.min() == col.max():
        return np.zeros(col.shape[0], dtype=np.int64)
    edges = np.asarray(np.percentile(col, np.linspace(0, 100, n_bins + 1)), dtype=np.float64)
    edges = edges[np.ediff1d(edges, to_begin=np.inf) > 1e-8]
    return np.searchsorted(edges[1:-1], col, side="right").astype(np.int64)


def _grouped_gap(group: np.ndarray, conf: np.ndarray, correct: np.ndarray) -> float:
    """sum over groups of |mean correct - mean conf| * count / N  =  sum_g |sum correct_g - sum conf_g| / N."""
    _, inv = np.unique(group, return_inverse=True)
    diff = np.bincount(inv, weights=correct) - np.bincount(inv, weights=conf)
    return float(np.abs(diff).sum() / conf.shape[0])


def AdaptiveECE(conf, pred, gt, conf_bin_num: int = 10) -> float:
    """Equal-mass-bin calibration error (tools/metrics.py:212-236)."""
    conf = np.asarray(conf)
    correct = (np.asarray(pred) == np.asarray(gt)).astype(np.float64)
    return _grouped_gap(quantile_bin_index(conf, conf_bin_num), conf.astype(np.float64), correct)


def PIECE(conf, knndist, pred, gt, dist_bin_num: int = 10, conf_bin_num: int = 10) -> float:
    """Proximity-informed ECE (tools/metrics.py:132-178, knn_strategy='quantile'): groups are (quantile bin of the
    proximity value, uniform confidence bin with conf == 1.0 kept in the last bin)."""
    conf = np.asarray(conf)
    correct = (np.asarray(pred) == np.asarray(gt)).astype(np.float64)
    knn_bin = quantile_bin_index(np.asarray(knndist), dist_bin_num)
    conf_bin = np.searchsorted(np.linspace(0, 1, conf_bin_num + 1)[1:-1], conf, side="right")
    return _grouped_gap(knn_bin * (conf_bin_num + 1) + conf_bin, conf.astype(np.float64), correct)


def macro_f1(pred, gt) -> float:
    """Unweighted mean of per-class F1 over the classes PRESENT in gt (vl_evaluator.py:77-82: sklearn f1_score with
    average='macro', labels=np.unique(labels)); a class never predicted scores 0."""
    pred, gt = np.asarray(pred).astype(np.int64), np.asarray(gt).astype(np.int64)
    classes = np.unique(gt)
    n = int(max(pred.max(), gt.max())) + 1
    tp = np.bincount(gt[pred == gt], minlength=n).astype(np.float64)
    n_pred = np.bincount(pred, minlength=n).astype(np.float64)
    n_gt = np.bincount(gt, minlength=n).astype(np.float64)
    denom = (n_pred + n_gt)[classes]                      # 2 tp + fp + fn
    f1 = np.where(denom > 0, 2 * tp[classes] / np.where(denom > 0, denom, 1), 0.0)
    return float(f1.mean())
