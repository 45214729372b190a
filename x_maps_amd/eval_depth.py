"""The reference's offline-evaluation caller of the hot path (python/eval/compute_depth_x_maps.py:81-114) on the GPU.

That script turns one camera time surface (a float image, 0 = no event) into "events" in raster order with a
normalised float timestamp, then calls the same per-event functions as the live pipe -- rectify, compute_event_disparity,
compute_disp_map_camera_view, disparity_to_depth_rectified -- and, for the point cloud, rectify_cam_coords_f32 +
construct_point_cloud.  Raster order means the timestamps are NOT sorted: this goes through the general
(min/max-reduction) path of the engine, never the time-sorted one.

File handling, the MC3D baseline and the metrics of eval/ are out of scope.
"""
from __future__ import annotations

import numpy as np

from .cam_proj_calibration import CamProjMaps
from .disp_to_depth import disparity_to_depth_rectified
from .x_maps_disparity import XMapsDisparity


def time_surface_to_events(cam_image):
    """eval/compute_depth_x_maps.py:83-97: normalise the non-zero times to [0, 1], clip negatives, list the pixels that
    stay > 0 in raster order.  (The earliest event lands on exactly 0 and is dropped -- as in the reference.)"""
    cam_image = np.array(cam_image, dtype=np.float64, copy=True)
    nz = cam_image != 0
    if not nz.any():
        return None
    lo, hi = cam_image[nz].min(), cam_image[nz].max()
    cam_image = (cam_image - lo) / (hi - lo)
    cam_image[cam_image < 0] = 0
    yx = np.argwhere(cam_image > 0)
    return {"x": yx[:, 1], "y": yx[:, 0], "t": cam_image[cam_image > 0]}


def compute_depth_from_time_surface(cam_proj_maps: CamProjMaps, x_maps_disp: XMapsDisparity, cam_image,
                                    want_point_cloud: bool = False, fused: bool = False):
    """-> (depth [cam_h][cam_w] float32, point_cloud [k][3] float32 or None); None, None for an empty surface.
    eval/compute_depth_x_maps.py:99-120.  fused=True runs the three fused kernels of the live path instead of the
    reference's stage sequence (needs cam_proj_maps built with camera_perspective=True); the depth map is identical."""
    events = time_surface_to_events(cam_image)
    if events is None:
        return None, None
    if fused and not want_point_cloud:
        if not cam_proj_maps.camera_perspective:
            raise ValueError("fused evaluation needs CamProjMaps(camera_perspective=True)")
        depth, _, _ = cam_proj_maps.engine.process_frame(events["x"], events["y"], events["t"], want_depth=True,
                                                         want_bgr=False)
        return depth, None
    xr, yr = cam_proj_maps.rectify_cam_coords_i16(events)
    disp, mask = x_maps_disp.compute_event_disparity(events=events, ev_x_rect_i16=xr, ev_y_rect_i16=yr)
    disparity = cam_proj_maps.compute_disp_map_camera_view(events=events, inlier_mask=mask, ev_disparity_f32=disp)
    depth = disparity_to_depth_rectified(disparity, cam_proj_maps.P2, engine=cam_proj_maps.engine)
    cloud = None
    if want_point_cloud:
        xr_f, yr_f = cam_proj_maps.rectify_cam_coords_f32(events)
        cloud = cam_proj_maps.construct_point_cloud(xr_f[mask], yr_f[mask], disp)
    return depth, cloud
