"""X-map construction on the GPU ("next" row N1): same call as the reference's
compute_x_map_from_time_map (python/x_map.py:5-55), which is a Numba prange kernel and the slowest step of
DepthReprojectionPipe.__post_init__ (python/depth_reprojection_pipe.py:85-90)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


def compute_x_map_from_time_map(time_map: np.ndarray, x_map_width: int, t_px_scale: int, X_OFFSET: int,
                                num_scanlines: int, device: int = 0):
    """(y, t) -> x + X_OFFSET by exhaustive per-row arg-min of |t - time_map[y, x]|; returns (x_map int16, t_diffs f32)."""
    tm = np.ascontiguousarray(time_map, dtype=np.float32)
    if tm.ndim != 2:
        raise ValueError("time_map must be 2-D")
    h, w = tm.shape
    x_map = np.empty((h, x_map_width), np.int16)
    t_diffs = np.empty((h, x_map_width), np.float32)
    lib = N.load_library()
    N.check(lib.xm_build_x_map(device, C.c_void_p(tm.ctypes.data), h, w, int(x_map_width), int(t_px_scale), int(X_OFFSET),
                               int(num_scanlines), C.c_void_p(x_map.ctypes.data), C.c_void_p(t_diffs.ctypes.data)))
    return x_map, t_diffs
