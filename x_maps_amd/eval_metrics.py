"""The evaluation metrics of the reference's table script (python/eval/create_evaluation_table.py:14-63) as a device
reduction (xm_eval_stats): fill rate, RMSE and the Middlebury-style error percentages of an estimated depth map against the
ground truth, optionally after load_and_filter.  The file handling of that script (globbing nine sequences, the MC3D / ESL
baselines, LaTeX output) is out of scope."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N


@dataclass
class EvaluationStats:
    fillrate: float
    rmse: float
    perc_1: float
    perc_5: float
    perc_10: float
    margin: float
    n_valid: int
    n_gt_zero: int


def evaluation_stats(estimate, groundtruth, min_depth=None, max_depth=None, device: int = 0) -> EvaluationStats:
    """evaluation_stats(estimate, groundtruth) (:14-54); with min_depth / max_depth given, load_and_filter (:57-62) is applied
    to the estimate first."""
    est = np.ascontiguousarray(estimate, dtype=np.float32)
    gt = np.ascontiguousarray(groundtruth, dtype=np.float32)
    if est.shape != gt.shape or est.ndim != 2:
        raise ValueError("estimate and groundtruth must be 2-D maps of the same shape")
    filt = min_depth is not None and max_depth is not None
    r = N.xm_eval_result()
    N.check(N.load_library().xm_eval_stats(device, C.c_void_p(est.ctypes.data), C.c_void_p(gt.ctypes.data), est.shape[0],
                                           est.shape[1], int(filt), float(min_depth or 0), float(max_depth or 0), C.byref(r)))
    return EvaluationStats(r.fillrate, r.rmse, r.perc_1, r.perc_5, r.perc_10, r.margin, int(r.n_valid), int(r.n_gt_zero))
