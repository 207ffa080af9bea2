"""kNN proximity (reference trainers/calibration/proximity.py) -- SURVEY §8(f) row f-2.

Same names and numpy-in / numpy-out contract as the reference; the per-query Python loop of ``torch.norm`` + ``topk`` is
one HIP kernel (``clipmi_knn_dists``).  ``*_device`` variants keep everything on the GPU."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import check, lib


def knn_dists_device(queries: torch.Tensor, refs: torch.Tensor, k: int) -> torch.Tensor:
    queries = ops._dev(queries, "queries", (torch.float32,))
    refs = ops._dev(refs, "refs", (torch.float32,))
    nq, e = queries.shape
    if refs.shape[1] != e:
        raise ValueError("knn: feature widths differ")
    out = torch.empty(nq, k, dtype=torch.float32, device=queries.device)
    check(lib.clipmi_knn_dists(queries.data_ptr(), refs.data_ptr(), out.data_ptr(), nq, refs.shape[0], e, k, ops._stream()),
          "clipmi_knn_dists")
    return out


def get_knn_dists(val_base_class_features, image_features_cur, K_nns):
    """proximity.py:19-46: distances of every current image feature to its K nearest base-val features, ascending."""
    q = torch.as_tensor(np.asarray(image_features_cur), dtype=torch.float32).cuda()
    r = torch.as_tensor(np.asarray(val_base_class_features), dtype=torch.float32).cuda()
    return knn_dists_device(q, r, K_nns).cpu().numpy()


def get_val_image_knn_dists(image_features_cur, K_nns):
    """proximity.py:49-70: K nearest OTHER rows of the same set (the zero self-distance is dropped)."""
    q = torch.as_tensor(np.asarray(image_features_cur), dtype=torch.float32).cuda()
    return knn_dists_device(q, q, K_nns + 1)[:, 1:].cpu().numpy()


def proximity_from_knn(knndists: np.ndarray) -> np.ndarray:
    """base_learner.py:136-137: exp(-mean distance to the K neighbours)."""
    return np.exp(-np.mean(knndists, axis=1))
