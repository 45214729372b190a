"""DisparityToDepth on the GPU (reference: python/disp_to_depth.py:66-115): 7x7 dilate composed with the
nearest remap to the projector view, and disparity -> depth -> u8 -> Turbo BGR with white 'no depth'."""
from __future__ import annotations

from dataclasses import dataclass

from .cam_proj_calibration import CamProjMaps


@dataclass
class DisparityToDepth:
    stats: object
    calib_maps: CamProjMaps
    z_near: float
    z_far: float

    def remap_rectified_disp_map_to_proj(self, rectified_disp_map):
        with self.stats.measure_time("remap disp"):  # dilate + remap are one kernel here
            return self.calib_maps.engine.remap_rectified_disp_map_to_proj(rectified_disp_map)

    def disparity_to_depth_rectified(self, disp_map):
        with self.stats.measure_time("d2d_rect"):
            return self.calib_maps.engine.disparity_to_depth(disp_map)

    def colorize_depth_from_disp(self, disp_map):
        with self.stats.measure_time("color_map"):  # d2d_rect + clip_norm + color_map fused
            return self.calib_maps.engine.colorize_depth_from_disp(disp_map)
