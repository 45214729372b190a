"""XMapsDisparity on the GPU: the reference's class (python/x_maps_disparity.py:35-82) owns the X-map and
answers compute_event_disparity(events, xr, yr) -> (disp[M] int16, inlier_mask[N] bool).  Here the X-map
lives in HBM inside the engine; the method runs K0 (extrema of t) + the A2 kernel and compacts on the host.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .cam_proj_calibration import CamProjMaps


@dataclass
class XMapsDisparity:
    cam_proj_maps: CamProjMaps

    X_OFFSET = 4242

    def __post_init__(self):
        self.proj_x_map = np.ascontiguousarray(self.cam_proj_maps.tables["proj_x_map"], dtype=np.int16)
        self.X_MAP_WIDTH = self.proj_x_map.shape[1]
        self.T_PX_SCALE = self.X_MAP_WIDTH - 1
        assert self.proj_x_map.shape[0] <= 2 ** 15 - 1  # int16 index headroom, as xmd:52-53

    def compute_event_disparity(self, events, ev_x_rect_i16, ev_y_rect_i16):
        disp_full, mask = self.cam_proj_maps.engine.event_disparity_full(ev_x_rect_i16, ev_y_rect_i16, events["t"])
        return disp_full[mask], mask
