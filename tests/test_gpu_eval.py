"""The evaluation caller of the hot path (eval/compute_depth_x_maps.py:81-114) and the point cloud on the GPU, against the
reference's golden vectors (g7) and the oracle."""
import os

import numpy as np
import pytest

import xmaps_oracle as O

pytestmark = pytest.mark.gpu


def _tables(g):
    return {"cam_mapx_i16": g["mapx"], "cam_mapy_i16": g["mapy"], "proj_x_map": g["xmap"],
            "rect_w": int(g["rect_w"]), "rect_h": int(g["rect_h"]), "p03": float(g["p03"]), "z_near": 0.1, "z_far": 1.0,
            "cam_mapx_f32": g["mapx_f32"], "cam_mapy_f32": g["mapy_f32"], "Q": g["Q"]}


@pytest.fixture(scope="module")
def g7(golden_dir):
    return np.load(os.path.join(golden_dir, "g7_eval_caller.npz"))


@pytest.mark.parametrize("fused", [False, True])
def test_eval_caller_golden(g7, fused):
    from x_maps_amd.cam_proj_calibration import CamProjMaps
    from x_maps_amd.eval_depth import compute_depth_from_time_surface, time_surface_to_events
    from x_maps_amd.x_maps_disparity import XMapsDisparity
    maps = CamProjMaps(_tables(g7), camera_perspective=True)
    xd = XMapsDisparity(maps)
    ev = time_surface_to_events(g7["raw_time_surface"])
    assert np.array_equal(ev["x"], g7["event_x"]) and np.array_equal(ev["t"], g7["event_t"])
    depth, cloud = compute_depth_from_time_surface(maps, xd, g7["raw_time_surface"], want_point_cloud=not fused, fused=fused)
    assert depth.dtype == np.float32 and depth.shape == g7["depth"].shape
    assert np.array_equal(depth == 0, g7["depth"] == 0)
    np.testing.assert_allclose(depth, g7["depth"], rtol=1e-4, atol=0)  # north_star tolerance on depth
    if not fused:
        # the float gather is bit-exact; the 4x4 transform is float32 arithmetic (BLAS sgemm in the reference): 1e-5 relative
        xf, yf = maps.rectify_cam_coords_f32(ev)
        assert np.array_equal(xf, g7["xr_f32"]) and np.array_equal(yf, g7["yr_f32"])
        ref = g7["cloud"]
        assert cloud.shape == ref.shape and cloud.dtype == np.float32
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(cloud), fin)
        np.testing.assert_allclose(cloud[fin], ref[fin], rtol=1e-5, atol=1e-6)


def test_empty_surface(g7):
    from x_maps_amd.cam_proj_calibration import CamProjMaps
    from x_maps_amd.eval_depth import compute_depth_from_time_surface
    from x_maps_amd.x_maps_disparity import XMapsDisparity
    maps = CamProjMaps(_tables(g7), camera_perspective=True)
    assert compute_depth_from_time_surface(maps, XMapsDisparity(maps), np.zeros((48, 64))) == (None, None)


def test_point_cloud_vs_oracle_large():
    from x_maps_amd.engine import XMapsEngine
    from x_maps_amd.synthetic import C_TINY, make_tables
    rng = np.random.default_rng(5)
    eng = XMapsEngine(make_tables(C_TINY))
    n = 300_000
    xp = rng.uniform(0, 1700, n).astype(np.float32)
    yp = rng.uniform(0, 1300, n).astype(np.float32)
    d = rng.integers(1, 600, n).astype(np.float32)
    Q = np.array([[1, 0, 0, -880.2], [0, 1, 0, -655.7], [0, 0, 0, 1234.5], [0, 0, -9.87, 0.031]])
    got = eng.construct_point_cloud(Q, xp, yp, d)
    np.testing.assert_allclose(got, O.construct_point_cloud(Q, xp, yp, d), rtol=1e-5, atol=1e-5)
    assert eng.construct_point_cloud(Q, xp[:0], yp[:0], d[:0]).shape == (0, 3)


def test_rectify_f32_index_error(g7):
    from x_maps_amd.cam_proj_calibration import CamProjMaps
    maps = CamProjMaps(_tables(g7), camera_perspective=True)
    with pytest.raises(IndexError):
        maps.rectify_cam_coords_f32({"x": np.array([1, 64]), "y": np.array([1, 2])})
    with pytest.raises(AttributeError):
        t = _tables(g7)
        del t["Q"]
        CamProjMaps(t, camera_perspective=True).construct_point_cloud(np.zeros(1, np.float32), np.zeros(1, np.float32), np.ones(1, np.float32))
