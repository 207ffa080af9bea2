"""Distance-Aware Calibration (reference trainers/calibration/distanse_aware_calibration.py).

``fit`` is a once-per-class-list host computation on four small [C,512] matrices and stays numpy (SURVEY a-11).
``predict`` keeps the reference's numpy-in / numpy-out contract but runs the row arg-max + scale on the GPU; in the
fused inference path the same scaling happens inside ``clipmi_logits`` (pass ``class_confidence_device``).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class DistanseAwareCalibration:
    def __init__(self):
        self.class_confidence = None
        self._dev = None

    def fit(self, base_text_features_zs, current_text_features_zs, base_text_features_tuned,
            current_text_features_tuned, k):
        """distanse_aware_calibration.py:13-46."""
        bz, cz = np.asarray(base_text_features_zs), np.asarray(current_text_features_zs)
        bt, ct = np.asarray(base_text_features_tuned), np.asarray(current_text_features_tuned)
        conf = np.empty(cz.shape[0], dtype=np.float64)
        for i in range(cz.shape[0]):
            d_zs = np.sort(np.linalg.norm(bz - cz[i], axis=1))[:k]
            d_fs = np.sort(np.linalg.norm(bt - ct[i], axis=1))[:k]
            zs_score = np.exp(-np.sum(d_zs) / k)
            fs_score = np.exp(-np.sum(d_fs) / k)
            conf[i] = 1.0 if d_fs[0] < 0.05 else fs_score / zs_score
        self.class_confidence = conf
        self._dev = None

    def class_confidence_device(self, device) -> torch.Tensor:
        if self._dev is None or self._dev.device != torch.device(device):
            self._dev = torch.from_numpy(self.class_confidence).float().to(device)
        return self._dev

    def predict(self, logits):
        """distanse_aware_calibration.py:49-58: ``logits[i] *= class_confidence[argmax_i]`` in fp32."""
        lg = torch.from_numpy(np.asarray(logits)).float().cuda()
        return scale_logits_(lg, self.class_confidence_device(lg.device)).cpu().numpy()


def scale_logits_(logits: torch.Tensor, class_confidence: torch.Tensor) -> torch.Tensor:
    """In-place DAC row scaling of an fp32 [N,C] device tensor through the row kernel of ``clipmi_logits``."""
    from ._lib import check, lib
    logits = ops._dev(logits, "logits", (torch.float32,))
    cc = ops._dev(class_confidence, "class_confidence", (torch.float32,))
    n, c = logits.shape
    check(lib.clipmi_calibrate_rows(logits.data_ptr(), cc.data_ptr(), None, None, n, c, ops._stream()), "clipmi_calibrate_rows")
    return logits
