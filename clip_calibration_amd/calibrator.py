"""VLCalibration on the branches the hot path covers (reference trainers/calibration/vl_calibrator.py:27-109, 170-200):
optional Distance-Aware Calibration of the logits, then softmax.  The binning / proximity base calibrators
(``base_calibration_mode`` 'scaling_based' / 'bin_based': netcal, isotonic, density-ratio) are outside SURVEY §8 and are
refused loudly rather than silently skipped."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .dac import DistanseAwareCalibration

TEXT_FEATURE_KEYS = ("base_text_features_zs", "current_text_features_zs", "base_text_features_tuned",
                     "current_text_features_tuned")


class VLCalibration:
    def __init__(self, val_dict: Dict[str, np.ndarray], text_feature_dict: Optional[Dict[str, np.ndarray]] = None,
                 dac_flag: bool = False, k_dac: int = 5, base_calibration_mode: Optional[str] = None, procal_flag: bool = False):
        if base_calibration_mode is not None or procal_flag:
            raise NotImplementedError("only the DAC / plain-softmax branches of VLCalibration are built (SURVEY §8 a-11, a-12)")
        self.dac_flag, self.k_dac = dac_flag, k_dac
        self.text_feature_dict = text_feature_dict
        self.val_logits = np.asarray(val_dict["val_logits"])
        self.val_labels = np.asarray(val_dict["val_labels"])
        self.val_image_features = np.asarray(val_dict["val_image_features"])
        self.val_image_knn_dists = np.asarray(val_dict["val_image_knn_dists"])
        self.val_image_proximity = np.exp(-np.mean(self.val_image_knn_dists, axis=-1))     # vl_calibrator.py:69
        self.dac_calibrator: Optional[DistanseAwareCalibration] = None

    def fit(self) -> None:
        """vl_calibrator.py:72-80 + build_dac_calibrator :170-200."""
        self.dac_calibrator = None
        if self.dac_flag:
            t = self.text_feature_dict
            if t is None or any(k not in t for k in TEXT_FEATURE_KEYS):
                raise KeyError(f"DAC needs text_feature_dict with {TEXT_FEATURE_KEYS}")
            self.dac_calibrator = DistanseAwareCalibration()
            self.dac_calibrator.fit(t["base_text_features_zs"], t["current_text_features_zs"],
                                    t["base_text_features_tuned"], t["current_text_features_tuned"], k=self.k_dac)

    def class_confidence_device(self, device="cuda") -> Optional[torch.Tensor]:
        """The per-class DAC factor as the fused logits kernel takes it (None when DAC is off)."""
        return None if self.dac_calibrator is None else self.dac_calibrator.class_confidence_device(device)

    def predict(self, logits, test_proximity=None) -> np.ndarray:
        """vl_calibrator.py:83-109 on the built branches: numpy [N,C] logits -> calibrated probabilities (float32: the DAC
        step already rounds to fp32 in the reference, and the row softmax runs in fp32 on the device)."""
        logits = np.asarray(logits)
        if test_proximity is not None and logits.shape[0] != np.asarray(test_proximity).shape[0]:
            raise AssertionError(f"Shape mismatch: logits shape {logits.shape[0]} != test_proximity shape {np.asarray(test_proximity).shape[0]}")
        lg = torch.from_numpy(logits).float().cuda()
        return ops.softmax_rows(lg, self.class_confidence_device(lg.device)).cpu().numpy()
