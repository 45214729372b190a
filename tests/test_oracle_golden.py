"""The CPU oracle (oracle/xmaps_oracle.py) against golden vectors captured from the reference's own
functions (tests/golden/make_golden.py).  Integer work: bit-exact.  Depth: exact (same FP64 divide)."""
import os

import numpy as np
import pytest

import xmaps_oracle as O

G1 = ["g1a_n1000", "g1b_n100000", "g1c_unsorted_dups", "g1d_rint_ties", "g1e_edges",
      "g1f_float32_t", "g1g_float64_t", "g1h_equal_t"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", G1)
def test_event_path_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    x, y, t = g["x"], g["y"], g["t"]
    S = int(g["t_px_scale"])
    xr, yr = O.rectify_cam_coords_i16(g["mapx"], g["mapy"], x, y)
    assert xr.dtype == np.int16 and np.array_equal(xr, g["xr"]) and np.array_equal(yr, g["yr"])
    disp, mask = O.compute_disparity(xr, yr, t, g["xmap"], S)
    assert disp.dtype == g["disp"].dtype == np.int16
    assert np.array_equal(mask, g["mask"])
    assert np.array_equal(disp, g["disp"])
    rh, rw, ch, cw = int(g["rect_h"]), int(g["rect_w"]), int(g["cam_h"]), int(g["cam_w"])
    if "proj_index_error" in g.files:
        with pytest.raises(IndexError):
            O.disp_map_projector_view(xr, yr, mask, disp, rh, rw)
    else:
        dm = O.disp_map_projector_view(xr, yr, mask, disp, rh, rw)
        assert dm.dtype == np.float32 and np.array_equal(dm, g["disp_map_proj"])
    dc = O.disp_map_camera_view(x, y, mask, disp, ch, cw)
    assert np.array_equal(dc, g["disp_map_cam"])


@pytest.mark.parametrize("name", [n for n in G1 if "float" not in n])
def test_key_frame_equals_last_writer_wins(golden_dir, name):
    """max over (event index, disp) keys == NumPy's fancy-assignment order, on the reference's output."""
    g = _load(golden_dir, name)
    tb = {"cam_mapx_i16": g["mapx"], "cam_mapy_i16": g["mapy"], "proj_x_map": g["xmap"],
          "t_px_scale": int(g["t_px_scale"]), "x_offset": 4242, "rect_h": int(g["rect_h"]),
          "rect_w": int(g["rect_w"]), "cam_h": int(g["cam_h"]), "cam_w": int(g["cam_w"])}
    t = g["t"]
    if "proj_index_error" not in g.files:
        kf = O.key_frame(tb, g["x"], g["y"], t, t.min(), t.max(), tag=7)
        assert np.array_equal(O.decode_key_frame(kf, tag=7), g["disp_map_proj"])
        assert not O.decode_key_frame(kf, tag=8).any()
    kc = O.key_frame(tb, g["x"], g["y"], t, t.min(), t.max(), tag=3, camera_perspective=True)
    assert np.array_equal(O.decode_key_frame(kc, tag=3), g["disp_map_cam"])


def test_edges_fixture_covers_the_cases(golden_dir):
    g = _load(golden_dir, "g1e_edges")
    yr = g["yr"]
    rh = int(g["rect_h"])
    assert {-1, 0, rh - 2, rh - 1} <= set(np.unique(yr).tolist())
    assert (g["disp"] == 0).any()
    # negative-column wrap was exercised: a cell in the last 12 columns is set by an event with xr < 0
    assert g["disp_map_proj"][9, int(g["rect_w"]) - 12] > 0
    d = _load(golden_dir, "g1d_rint_ties")
    S = int(d["t_px_scale"])
    tn = (d["t"] - d["t"].min()) / (d["t"].max() - d["t"].min()) * S
    assert (np.abs(tn - np.floor(tn) - 0.5) < 1e-12).sum() >= 60  # exact .5 ties present


def test_frame_stages_match_reference(golden_dir):
    g = _load(golden_dir, "g2_frame_stages")
    depth = O.disparity_to_depth_rectified(g["disp"], float(g["p03"]))
    assert depth.dtype == np.float32 and np.array_equal(depth, g["depth"])
    assert np.array_equal(O.disparity_to_depth_rectified(g["disp"], float(g["p03_neg"])), g["depth_neg"])
    # golden comes from the stub run (f32 product under NumPy 2) -> compare that variant bit-exact,
    # and require the Numba-typed variant (f64 product) to differ by at most 1 LSB
    u8_f32 = O.clip_normalize_uint8_depth_frame(depth, float(g["z_near"]), float(g["z_far"]), mul_in_f64=False)
    assert np.array_equal(u8_f32, g["u8"])
    u8_f64 = O.clip_normalize_uint8_depth_frame(depth, float(g["z_near"]), float(g["z_far"]), mul_in_f64=True)
    assert np.abs(u8_f64.astype(int) - g["u8"].astype(int)).max() <= 1
    white = g["frame_in"].copy()
    white[g["u8"] == 0] = 255
    assert np.array_equal(white, g["frame_white"])
    bgr = O.generate_color_map(g["u8"])
    assert bgr.shape == (32, 32, 3) and (bgr[g["u8"] == 0] == 255).all()


def test_x_map_builder_matches_reference(golden_dir):
    g = _load(golden_dir, "g3_x_map")
    xm, td = O.compute_x_map_from_time_map(g["time_map"], int(g["x_map_width"]), int(g["t_px_scale"]), 4242,
                                           int(g["num_scanlines"]))
    assert np.array_equal(xm, g["x_map"])
    # t_diffs is a by-product the reference throws away (xmd:61).  Under the stub run `t - t_map` is
    # python-float minus np.float32 = float32 (NEP 50); Numba evaluates it in float64 (what the oracle
    # does), so this output only agrees to f32 rounding.
    assert np.allclose(td, g["t_diffs"], rtol=0, atol=2e-7)
    assert (xm[:, 0] == 0).all() and (xm > 0).any()


def test_linear_time_map_matches_reference(golden_dir):
    g = _load(golden_dir, "g4_time_map")
    for key in g.files:
        _, wh, d = key.split("_")
        w, h = map(int, wh.split("x"))
        assert np.array_equal(O.generate_linear_projector_time_map(w, h, d == "up"), g[key])


def test_dilate_and_remap_properties():
    """A4 is unpinned (OpenCV absent): check the restated semantics against a brute-force definition."""
    rng = np.random.default_rng(5)
    f = rng.integers(0, 50, (23, 31)).astype(np.float32)
    f[rng.random(f.shape) < 0.7] = 0
    d = O.dilate7x7(f)
    brute = np.zeros_like(f)
    for i in range(f.shape[0]):
        for j in range(f.shape[1]):
            brute[i, j] = f[max(0, i - 3):i + 4, max(0, j - 3):j + 4].max()
    assert np.array_equal(d, brute)
    m = np.stack((rng.integers(-3, 35, (9, 11)), rng.integers(-3, 27, (9, 11))), -1).astype(np.int16)
    r = O.remap_nearest_i16(d, m)
    for v in range(9):
        for u in range(11):
            mx, my = m[v, u]
            exp = d[my, mx] if (0 <= mx < 31 and 0 <= my < 23) else 0
            assert r[v, u] == exp


def test_eval_caller_matches_reference(golden_dir):
    """The offline-evaluation caller (eval/compute_depth_x_maps.py:81-114) and construct_point_cloud, restated."""
    g = _load(golden_dir, "g7_eval_caller")
    x, y, t = O.time_surface_to_events(g["raw_time_surface"])
    assert np.array_equal(x, g["event_x"]) and np.array_equal(y, g["event_y"]) and np.array_equal(t, g["event_t"])
    assert np.any(np.diff(t) < 0)  # raster order: not time-sorted
    xr, yr = O.rectify_cam_coords_i16(g["mapx"], g["mapy"], x, y)
    disp, mask = O.compute_disparity(xr, yr, t, g["xmap"], int(g["t_px_scale"]))
    assert np.array_equal(disp, g["disp"]) and np.array_equal(mask, g["mask"])
    ch, cw = g["mapx"].shape
    dm = O.disp_map_camera_view(x, y, mask, disp, ch, cw)
    assert np.array_equal(dm, g["disp_map"])
    assert np.array_equal(O.disparity_to_depth_rectified(dm, float(g["p03"])), g["depth"])
    xf, yf = O.rectify_cam_coords_f32(g["mapx_f32"], g["mapy_f32"], x, y)
    assert xf.dtype == np.float32 and np.array_equal(xf, g["xr_f32"]) and np.array_equal(yf, g["yr_f32"])
    cloud = O.construct_point_cloud(g["Q"], xf[mask], yf[mask], disp)
    assert cloud.dtype == np.float32 and cloud.shape == g["cloud"].shape
    assert np.array_equal(cloud, g["cloud"], equal_nan=True)
    assert np.isfinite(cloud).all() == bool((disp != 0).all())


def test_eval_metrics_oracle_matches_reference(golden_dir):
    """G8: class evaluation_stats + load_and_filter of python/eval/create_evaluation_table.py."""
    g = np.load(os.path.join(golden_dir, "g8_eval_metrics.npz"))
    for k in "abc":
        est = O.load_and_filter(g[f"{k}_est_raw"], g[f"{k}_gt"], float(g["min_depth"]), float(g["max_depth"]))
        assert np.array_equal(est, g[f"{k}_est"])
        r = O.evaluation_stats(est, g[f"{k}_gt"])
        got = np.array([r["fillrate"], r["rmse"], r["perc_1"], r["perc_5"], r["perc_10"], r["margin"]])
        np.testing.assert_allclose(got, g[f"{k}_res"], rtol=1e-12, atol=0)
    r = O.evaluation_stats(np.zeros_like(g["a_gt"]), g["a_gt"])
    assert r["rmse"] == 0 and r["fillrate"] == g["empty_res"][0]


def test_a4_dilation_equals_an_independent_maximum_filter():
    """A4 is restated from OpenCV's documentation (no cv2 offline).  An independent implementation of the same definition --
    SciPy's 7 x 7 maximum filter with a constant border that can never win (cv2.dilate's default border value is the type's
    minimum, python/disp_to_depth.py:84-85 runs it on non-negative disparities) -- followed by the integer gather that
    cv2.remap(INTER_NEAREST) with a CV_16SC2 map and BORDER_CONSTANT 0 performs (:88-96) must give the same frames."""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(17)
    for h, w in ((40, 56), (7, 9), (3, 3), (64, 1), (1, 64)):
        rect = np.where(rng.random((h, w)) < 0.15, rng.integers(1, 300, (h, w)), 0).astype(np.float32)
        assert np.array_equal(O.dilate7x7(rect), ndi.maximum_filter(rect, size=7, mode="constant", cval=0.0))
        ph, pw = 23, 31
        mxy = np.stack((rng.integers(-4, w + 4, (ph, pw)), rng.integers(-4, h + 4, (ph, pw))), axis=-1).astype(np.int16)
        dil = ndi.maximum_filter(rect, size=7, mode="constant", cval=0.0)
        mx, my = mxy[..., 0].astype(np.int64), mxy[..., 1].astype(np.int64)
        inside = (mx >= 0) & (mx < w) & (my >= 0) & (my < h)
        want = np.where(inside, dil[np.clip(my, 0, h - 1), np.clip(mx, 0, w - 1)], np.float32(0))
        assert np.array_equal(O.remap_rectified_disp_map_to_proj(rect, mxy), want)
