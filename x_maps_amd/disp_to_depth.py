"""DisparityToDepth on the GPU (reference: python/disp_to_depth.py:66-115): 7x7 dilate composed with the
nearest remap to the projector view, and disparity -> depth -> u8 -> Turbo BGR with white 'no depth'."""
from __future__ import annotations

from dataclasses import dataclass

from .cam_proj_calibration import CamProjMaps


def disparity_to_depth_rectified(disparity, P2, engine):
    """Module-level form the evaluation script calls (python/disp_to_depth.py:46-63): depth = max(P2[0,3]/d, 1e-9)
    where d != 0.  `engine` is the XMapsEngine whose p03 must be P2[0,3] (it is a table of that engine)."""
    if abs(float(P2[0, 3]) - engine.p03) > 0:
        raise ValueError("P2[0,3] differs from the engine's p03")
    return engine.disparity_to_depth(disparity)


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
