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


def test_eval_metrics_on_the_gpu_match_the_reference(golden_dir):
    """G8: evaluation_stats / load_and_filter of python/eval/create_evaluation_table.py as a device reduction.  Counts are
    exact up to pixels sitting on the margin (f64 sums here, f32 pairwise sums there): <= 2 pixels; RMSE / margin 1e-6."""
    from x_maps_amd.eval_metrics import evaluation_stats
    g = np.load(os.path.join(golden_dir, "g8_eval_metrics.npz"))
    lo, hi = float(g["min_depth"]), float(g["max_depth"])
    for k in "abc":
        gt, want = g[f"{k}_gt"], g[f"{k}_res"]
        for est, kw in ((g[f"{k}_est"], {}), (g[f"{k}_est_raw"], {"min_depth": lo, "max_depth": hi})):
            r = evaluation_stats(est, gt, **kw)
            px = gt.size
            assert abs(r.fillrate - want[0]) <= 2.0 / max(px - r.n_gt_zero, 1)
            np.testing.assert_allclose([r.rmse, r.margin], [want[1], want[5]], rtol=1e-6)
            np.testing.assert_allclose([r.perc_1, r.perc_5, r.perc_10], want[2:5], rtol=0, atol=100.0 * 1 / px)
            ref = O.evaluation_stats(O.load_and_filter(est, gt, lo, hi) if kw else est, gt)
            np.testing.assert_allclose([r.rmse, r.perc_1, r.perc_5, r.perc_10], [ref["rmse"], ref["perc_1"], ref["perc_5"], ref["perc_10"]],
                                       rtol=1e-6, atol=100.0 / px)
    r = evaluation_stats(np.zeros_like(g["a_gt"]), g["a_gt"])
    assert r.rmse == 0 and r.n_valid == 0 and r.fillrate == g["empty_res"][0]


def _write_esl_yaml(path, g, cam_D):
    def node(name, a):
        a = np.asarray(a, dtype=float)
        a = a.reshape(a.shape[0], -1)
        data = ", ".join(repr(float(v)) for v in a.ravel())
        return f"{name}: !!opencv-matrix\n   rows: {a.shape[0]}\n   cols: {a.shape[1]}\n   dt: d\n   data: [ {data} ]\n"
    txt = "%YAML:1.0\n---\n" + node("cam_K", g["camera_K"]) + node("cam_kc", np.reshape(cam_D, (1, 5))) + \
          node("proj_K", g["projector_K"]) + node("proj_kc", g["projector_D"]) + node("R", g["R"]) + node("T", g["T"])
    path.write_text(txt)


def test_esl_yaml_evaluation_configuration(tmp_path, golden_dir):
    """CamProjCalibrationParams.from_ESL_yaml + the evaluation's table configuration (python/cam_proj_calibration.py:110-140,
    python/eval/compute_depth_x_maps.py:57-77: rect = 3 x projector = 3240 x 5760, zero_undistort_proj_map, scan_upwards=False,
    BORDER_CONSTANT) and the evaluation caller on those tables: a raster-order time surface through the camera-view engine
    (X-map 5760 rows tall: the LDS window shrinks or the direct kernel takes over) == oracle."""
    from x_maps_amd import calibration as C
    from x_maps_amd import rig
    from x_maps_amd.cam_proj_calibration import CamProjMaps
    from x_maps_amd.eval_depth import compute_depth_from_time_surface, time_surface_to_events
    from x_maps_amd.x_maps_disparity import XMapsDisparity
    g = np.load(os.path.join(golden_dir, "g6_esl_calib.npz"))
    ypath = tmp_path / "ESL_calib.yaml"
    _write_esl_yaml(ypath, g, rig.NEBRA_CAMERA_D)
    cp = C.CamProjCalibrationParams.from_ESL_yaml(str(ypath), 640, 480, 1080, 1920)
    assert (cp.rect_image_width, cp.rect_image_height) == (3240, 5760)
    assert np.array_equal(cp.camera_K, g["camera_K"]) and np.array_equal(cp.projector_D.ravel(), g["projector_D"].ravel())
    tb = C.build_eval_tables(cp)
    assert tb["proj_x_map"].shape == (5760, 1080) and tb["cam_mapx_f32"].shape == (480, 640)
    # the X-map the GPU built == the oracle's construction on a band of rows (the full 5760 x 1080 x 3240 search is slow on the CPU)
    rows = slice(2800, 2830)
    xm, _ = O.compute_x_map_from_time_map(tb["time_map_rect"][rows], tb["x_map_width"], tb["t_px_scale"], 4242, cp.projector_width)
    assert np.array_equal(xm, tb["proj_x_map"][rows])
    # scan_upwards=False + BORDER_CONSTANT: outside the projector image the rectified time map is 0 (undefined), not replicated
    assert (tb["time_map_rect"] == 0).mean() > 0.2
    # a time surface: the scene rendered by the rig with the projector scanning downwards, last time stamp per pixel
    evs, _ = rig.render_events(cp, tb, row_stride=7, scan_upwards=False)
    surf = np.zeros((480, 640), np.float32)
    surf[evs["y"], evs["x"]] = (evs["t"] - evs["t"].min() + 1).astype(np.float32)
    maps = CamProjMaps(tb, camera_perspective=True)
    xd = XMapsDisparity(maps)
    depth, cloud = compute_depth_from_time_surface(maps, xd, surf, want_point_cloud=True)
    depth_fused, _ = compute_depth_from_time_surface(maps, xd, surf, fused=True)
    ev = time_surface_to_events(surf)
    ref = O.process_ev_frame(tb, ev["x"].astype(np.int64), ev["y"].astype(np.int64), ev["t"], camera_perspective=True, want_bgr=False)
    assert ref["mask"].mean() > 0.5
    assert np.array_equal(depth == 0, ref["depth"] == 0) and np.array_equal(depth_fused, depth)
    np.testing.assert_allclose(depth, ref["depth"], rtol=1e-4, atol=0)
    assert cloud.shape == (int(ref["mask"].sum()), 3) and np.isfinite(cloud).all()
    # and the metrics close the loop: the depth map against itself is perfect
    from x_maps_amd.eval_metrics import evaluation_stats
    st = evaluation_stats(depth * 100, depth * 100)
    assert st.rmse == 0 and st.fillrate == 1.0
