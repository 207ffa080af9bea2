"""Checkpoint and cache formats either side of the path -- SURVEY §8(f) row f-3.  Host-only.

* Dassl prompt-learner checkpoints ``<dir>/<name>/model.pth.tar-<epoch>`` / ``model-best.pth.tar`` as the reference's
  ``load_model`` reads them (trainers/classification/coop.py:311-343, same in kgcoop.py / maple.py): a pickled dict with
  ``state_dict`` and ``epoch``; the fixed ``token_prefix`` / ``token_suffix`` buffers are dropped before a non-strict load.
* ``base_features.pt``: the dict ``save_base_val_features`` writes (base_learner.py:184-239) and ``test()`` /
  ``get_text_features`` read back (base_learner.py:110-113, 243-252).
"""
from __future__ import annotations

import os
import os.path as osp
from typing import Dict, Optional

import numpy as np
import torch

BASE_FEATURE_KEYS = ("val_logits", "val_image_features", "val_text_features", "val_labels", "val_image_knn_dists")


def checkpoint_path(directory: str, name: str, epoch: Optional[int] = None) -> str:
    return osp.join(directory, name, "model-best.pth.tar" if epoch is None else f"model.pth.tar-{epoch}")


def load_checkpoint(path: str) -> Dict:
    if not osp.exists(path):
        raise FileNotFoundError(f'Model not found at "{path}"')
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if "state_dict" not in ckpt:
        raise KeyError(f'"{path}" holds no state_dict')
    return ckpt


def save_checkpoint(state_dict: Dict[str, torch.Tensor], directory: str, name: str, epoch: int, **extra) -> str:
    """Writes the layout ``load_model`` expects (Dassl's save_checkpoint: ``model.pth.tar-<epoch>`` plus a ``checkpoint``
    file naming it)."""
    path = checkpoint_path(directory, name, epoch)
    os.makedirs(osp.dirname(path), exist_ok=True)
    torch.save({"state_dict": {k: v.detach().cpu() for k, v in state_dict.items()}, "epoch": epoch, **extra}, path)
    with open(osp.join(osp.dirname(path), "checkpoint"), "w") as f:
        f.write(osp.basename(path) + "\n")
    return path


def load_model(module: torch.nn.Module, directory: str, name: str = "prompt_learner", epoch: Optional[int] = None) -> int:
    """coop.py:311-343 for one named sub-module: returns the checkpoint's epoch."""
    ckpt = load_checkpoint(checkpoint_path(directory, name, epoch))
    state_dict = dict(ckpt["state_dict"])
    for fixed in ("token_prefix", "token_suffix", "prompt_learner.token_prefix", "prompt_learner.token_suffix"):
        state_dict.pop(fixed, None)
    module.load_state_dict(state_dict, strict=False)
    return int(ckpt.get("epoch", -1))


def base_features_path(root: str, dataset: str, trainer: str, shots: int, backbone: str, seed: int) -> str:
    """./temp/base_features/<dataset>/<trainer>/shots<k>/<backbone>/base/seed<s>/base_features.pt (base_learner.py:110-112)."""
    return osp.join(root, dataset, trainer, f"shots{shots}", backbone, "base", f"seed{seed}", "base_features.pt")


def save_base_features(path: str, val_logits, val_image_features, val_text_features, val_labels, val_image_knn_dists) -> None:
    os.makedirs(osp.dirname(path) or ".", exist_ok=True)
    torch.save({"val_logits": np.asarray(val_logits), "val_image_features": np.asarray(val_image_features),
                "val_text_features": np.asarray(val_text_features), "val_labels": np.asarray(val_labels),
                "val_image_knn_dists": np.asarray(val_image_knn_dists)}, path)


def load_base_features(path: str) -> Dict[str, np.ndarray]:
    d = torch.load(path, map_location="cpu", weights_only=False)
    missing = [k for k in BASE_FEATURE_KEYS if k not in d]
    if missing:
        raise KeyError(f'"{path}" lacks {missing}')
    return {k: np.asarray(d[k]) for k in BASE_FEATURE_KEYS}
