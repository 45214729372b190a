"""Runtime half of the reference's cam_proj_calibration.py on the GPU.

`CamProjMaps` here holds the already-built int16 tables (what CamProjMaps.__post_init__ produces with
OpenCV at python/cam_proj_calibration.py:174-270 -- that setup step needs cv2 and is out of scope, see
DESIGN.md) and exposes the three per-frame methods with the reference's names and arguments
(python/cam_proj_calibration.py:277-281, 299-303, 312-317).  Each one is a HIP kernel behind the C-ABI.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from types import SimpleNamespace

import numpy as np

from .engine import XMapsEngine

TABLE_KEYS = ("cam_mapx_i16", "cam_mapy_i16", "proj_x_map", "disp_proj_mapxy_i16", "rect_w", "rect_h", "p03",
              "z_near", "z_far")


def load_tables_npz(path: str) -> dict:
    """Tables exported once from a reference installation (INTEGRATION.md has the 10-line exporter)."""
    with np.load(path) as z:
        tb = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
    for k in ("cam_mapx_i16", "cam_mapy_i16", "proj_x_map", "rect_w", "rect_h", "p03"):
        if k not in tb:
            raise KeyError(f"{path}: missing table '{k}'")
    tb.setdefault("x_offset", 4242)
    return tb


def _events_xy(events):
    return events["x"], events["y"]


@dataclass
class CamProjMaps:
    """Tables + the GPU engine that consumes them.  `calib` mirrors the attribute names the reference's
    methods read (rect_image_{width,height}, camera_{width,height}, projector_{width,height})."""
    tables: dict
    camera_perspective: bool = False
    device: int = 0
    n_slots: int = 1
    assume_time_sorted: bool = False
    engine: XMapsEngine = field(init=False)

    def __post_init__(self):
        tb = self.tables
        self.disp_cam_mapx_i16 = np.ascontiguousarray(tb["cam_mapx_i16"], dtype=np.int16)
        self.disp_cam_mapy_i16 = np.ascontiguousarray(tb["cam_mapy_i16"], dtype=np.int16)
        self.disp_proj_mapxy_i16 = tb.get("disp_proj_mapxy_i16")
        cam_h, cam_w = self.disp_cam_mapx_i16.shape
        proj_h, proj_w = self.disp_proj_mapxy_i16.shape[:2] if self.disp_proj_mapxy_i16 is not None else (0, 0)
        self.calib = SimpleNamespace(rect_image_width=int(tb["rect_w"]), rect_image_height=int(tb["rect_h"]),
                                     camera_width=cam_w, camera_height=cam_h, projector_width=proj_w,
                                     projector_height=proj_h)
        self.P2 = np.zeros((3, 4))
        self.P2[0, 3] = float(tb["p03"])
        # optional: float rectify maps and Q, only the offline evaluation caller needs them
        self.disp_cam_mapx_f32 = tb.get("cam_mapx_f32")
        self.disp_cam_mapy_f32 = tb.get("cam_mapy_f32")
        self.Q = tb.get("Q")
        self.engine = XMapsEngine(tb, camera_perspective=self.camera_perspective, device=self.device,
                                  n_slots=self.n_slots, assume_time_sorted=self.assume_time_sorted)

    def rectify_cam_coords_i16(self, events):
        x, y = _events_xy(events)
        return self.engine.rectify_cam_coords_i16(x, y)

    def rectify_cam_coords_f32(self, events):
        """python/cam_proj_calibration.py:272-275"""
        if self.disp_cam_mapx_f32 is None or self.disp_cam_mapy_f32 is None:
            raise AttributeError("tables hold no cam_mapx_f32 / cam_mapy_f32 (export them to use the f32 rectification)")
        x, y = _events_xy(events)
        return self.engine.rectify_cam_coords_f32(self.disp_cam_mapx_f32, self.disp_cam_mapy_f32, x, y)

    def construct_point_cloud(self, xpr_f32, ypr_f32, disp_f32):
        """python/cam_proj_calibration.py:319-331"""
        if self.Q is None:
            raise AttributeError("tables hold no Q (4x4 reprojection matrix)")
        return self.engine.construct_point_cloud(self.Q, xpr_f32, ypr_f32, disp_f32)

    def compute_disp_map_projector_view(self, ev_x_rect_i16, ev_y_rect_i16, inlier_mask, ev_disparity_f32):
        full = np.zeros(len(inlier_mask), np.int16)
        full[inlier_mask] = ev_disparity_f32  # the reference passes the compacted int16 disparities
        return self.engine.disp_map_projector_view(ev_x_rect_i16, ev_y_rect_i16, full, inlier_mask)

    def compute_disp_map_camera_view(self, events, inlier_mask, ev_disparity_f32):
        x, y = _events_xy(events)
        full = np.zeros(len(inlier_mask), np.int16)
        full[inlier_mask] = ev_disparity_f32
        return self.engine.disp_map_camera_view(x, y, full, inlier_mask)
