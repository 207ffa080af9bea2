"""The callers either side of the hot path: VLBaseLearner.test / save_base_val_features / get_text_features
(reference trainers/classification/base_learner.py:59-152, 184-300) -- SURVEY §8(f) rows f-1..f-3, restated around the
device-side path.  What changes against the reference's loop:

* host batches reach the GPU on a side copy stream, a batch ahead (``device_batches``: ``non_blocking`` from pinned memory), instead
  of a synchronous ``.to(device)`` on the compute stream per batch;
* logits, features and labels stay on the GPU; per batch the fused logits kernel already applies DAC and returns
  (conf, pred), which feed the device ECE accumulators -- no ``.cpu().numpy().tolist()`` per batch;
* test-image proximity is one kNN kernel launch over the kept [N,E] features instead of a Python loop per query;
* everything that needs the samples (macro-F1, ACE, PIECE) runs once on 16 B per sample.

``infer`` is any of the trainer mirrors' inference callables: ``ZeroshotCLIP.model_inference``, ``CustomCLIP.__call__``,
``CustomCLIPCalibration.__call__`` -- signature ``(image, dac_conf=None, want_conf_pred=False)``.
A ``loader`` is any iterable of ``(image[B,3,R,R], label[B])`` (Dassl's ``parse_batch_test`` output, base_learner.py:175-182).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from .calibrator import VLCalibration
from .evaluator import DeviceCalibrationEvaluator
from .proximity import knn_dists_device


def device_batches(loader: Iterable[Tuple[torch.Tensor, torch.Tensor]], device="cuda", depth: int = 2):
    """``parse_batch_test`` for a host loader (Dassl: ``batch["img"].to(device), batch["label"].to(device)``, reference
    trainers/classification/base_learner.py:84-88,175-182) without its per-batch stall: the copy of batch i + 1 is issued on a SIDE
    stream while batch i computes, and the consumer's stream waits on the copy's event, never the host.
    * pinned host tensors (a DataLoader with ``pin_memory=True``) are copied ``non_blocking``: the host thread does not wait either;
    * pageable tensors go through torch's own staged copy on the side stream -- the host blocks for the copy, but by then it has
      queued the previous batch's launches, so the GPU computes meanwhile (an extra host-side copy into a pinned ring was measured
      3x SLOWER than that: 154 MB of memcpy per batch of 256 on one host thread, profiles/r03_stream_input.txt);
    * tensors that already live on the device pass through.
    Yields (image_on_device, label_on_device), ``depth - 1`` batches ahead."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("device_batches: the path runs on a ROCm GPU only")
    copy_stream = torch.cuda.Stream(device=dev)
    pending = []                    # (device image, device label, copy-done event), oldest first

    def stage(image, label):
        image, label = torch.as_tensor(image), torch.as_tensor(label)
        if image.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            return image, label.to(dev, non_blocking=True), ev
        with torch.cuda.stream(copy_stream):
            d_img = image.to(dev, non_blocking=image.is_pinned())
            d_lab = label.to(dev, non_blocking=label.is_pinned())
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        return d_img, d_lab, ev

    def release(item):
        d_img, d_lab, ev = item
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(ev)                                  # device-side wait: the host runs on
        d_img.record_stream(cur)                            # allocated on the copy stream, consumed on this one
        d_lab.record_stream(cur)
        return d_img, d_lab

    for image, label in loader:
        pending.append(stage(image, label))
        if len(pending) >= depth:
            yield release(pending.pop(0))
    while pending:
        yield release(pending.pop(0))


def _call(infer: Callable, image: torch.Tensor, dac_conf=None, want_conf_pred=False):
    try:
        return infer(image, dac_conf=dac_conf, want_conf_pred=want_conf_pred)
    except TypeError as e:          # a reference-style 3-tuple callable without the fused extras
        if dac_conf is not None or want_conf_pred:
            raise TypeError("infer must accept dac_conf= and want_conf_pred= (use the clip_calibration_amd.trainers mirrors)") from e
        return infer(image)


@torch.no_grad()
def collect_base_val_features(infer: Callable, loader: Iterable[Tuple[torch.Tensor, torch.Tensor]], image_k: int = 10,
                              device="cuda") -> Dict[str, np.ndarray]:
    """save_base_val_features (base_learner.py:184-239) minus the torch.save: one pass over the base-class val split with
    the current model; returns the dict the reference stores as base_features.pt (see checkpoint.save_base_features)."""
    logits, feats, labels, text = [], [], [], None
    for image, label in device_batches(loader, device):
        out = _call(infer, image)
        logits.append(out[0])
        feats.append(out[1])
        labels.append(label)
        text = out[2]
    if text is None:
        raise ValueError("empty loader")
    feats_d = torch.cat(feats).float()
    k = min(image_k, feats_d.shape[0] - 1)
    knn = knn_dists_device(feats_d, feats_d, k + 1)[:, 1:] if k > 0 else feats_d.new_zeros(feats_d.shape[0], 0)
    return {"val_logits": torch.cat(logits).float().cpu().numpy(), "val_image_features": feats_d.cpu().numpy(),
            "val_text_features": text.float().cpu().numpy(), "val_labels": torch.cat(labels).cpu().numpy(),
            "val_image_knn_dists": knn.cpu().numpy()}


def text_feature_dict(base_zs: Dict[str, np.ndarray], current_text_features_zs, base_tuned: Dict[str, np.ndarray],
                      current_text_features_tuned) -> Dict[str, np.ndarray]:
    """get_text_features (base_learner.py:241-300): the four matrices DAC.fit consumes.  ``base_zs`` / ``base_tuned`` are
    the base_features.pt dicts of the zero-shot and of the tuned model; the two ``current_*`` are the L2-normalised text
    features of the classes under test from ``ZeroshotCLIP`` and from the tuned trainer (its 3-tuple's last element)."""
    as_np = lambda t: t.float().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return {"base_text_features_zs": np.asarray(base_zs["val_text_features"]),
            "current_text_features_zs": as_np(current_text_features_zs),
            "base_text_features_tuned": np.asarray(base_tuned["val_text_features"]),
            "current_text_features_tuned": as_np(current_text_features_tuned)}


@torch.no_grad()
def test(infer: Callable, loader: Iterable[Tuple[torch.Tensor, torch.Tensor]], val_dict: Optional[Dict[str, np.ndarray]] = None,
         calibrator: Optional[VLCalibration] = None, image_k: int = 10, ece_bins: int = 10, piece_bins: int = 10,
         device="cuda", group=None) -> "OrderedDict[str, float]":
    """VLBaseLearner.test (base_learner.py:59-152): inference over the split, DAC, softmax top-1, proximity of every test
    image to the base-class val images (exp(-mean K-NN distance), :121-137), then the evaluator's metrics.  Under
    torch.distributed each rank passes its own shard of the loader; samples are gathered before the sample-level metrics."""
    ev = DeviceCalibrationEvaluator(ece_bins, device=device, keep_samples=True, piece_bins=piece_bins)
    dac = calibrator.class_confidence_device(device) if calibrator is not None else None
    feats = []
    for image, label in device_batches(loader, device):
        out = _call(infer, image, dac_conf=dac, want_conf_pred=True)
        ev.process(out[3], out[4], label)
        feats.append(out[1])
    proximity = None
    if val_dict is not None and feats:
        refs = torch.as_tensor(np.asarray(val_dict["val_image_features"]), dtype=torch.float32, device=device)
        k = min(image_k, refs.shape[0])
        proximity = torch.exp(-knn_dists_device(torch.cat(feats).float(), refs, k).mean(dim=1))
    if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                             and torch.distributed.get_world_size() > 1):
        from .parallel import gather_samples
        proximity = gather_samples(ev, proximity, group, has_proximity=val_dict is not None)   # rank-uniform, not data-dependent
    return ev.evaluate(None if proximity is None else proximity.cpu().numpy())
