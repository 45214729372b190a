"""Ideal projector time map (reference python/proj_time_map.py:6-19): the projector draws x-slow, y-fast; a tiny
host-side table (proj_w x proj_h floats), restated in NumPy.  Its rectification (cv2.remap, :22-29) belongs to the
OpenCV-based calibration setup and is out of scope."""
from __future__ import annotations

import numpy as np


def generate_linear_projector_time_map(proj_width: int, proj_height: int, scan_upwards: bool) -> np.ndarray:
    rows = np.arange(proj_height)[:, None]
    cols = np.arange(proj_width)[None, :]
    if scan_upwards:
        rows = rows[::-1]
    order = cols * proj_height + rows  # column-major pixel order = drawing order
    return (order / (proj_width * proj_height)).astype(np.float32)
