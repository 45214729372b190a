"""-m gpu: the fused HIP path (through the C-ABI) against the CPU oracle and the golden vectors."""
import os

import numpy as np
import pytest

from conftest import xm_option

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu

REL_TOL_DEPTH = 1e-4  # BASELINE.json north_star: float depth within 1e-4 relative (we expect 0)


def _check_frame(tb, evs, camera, p=None, shuffled=False):
    x, y, t, pp = S.to_soa(evs)
    use = np.ones(len(evs), bool) if p is None else (pp == 1)
    ref = O.process_ev_frame(tb, x[use].astype(np.int64), y[use].astype(np.int64), t[use], camera_perspective=camera)
    with XMapsEngine(tb, camera_perspective=camera) as eng:
        depth, bgr, st = eng.process_frame(x, y, t, pp if p is not None else None)
        dbg = eng.debug_event_outputs(x, y, t, pp if p is not None else None)
        depth2, bgr2, st2 = eng.process_events(evs, use_polarity=p is not None)
    # integer index path: bit-exact per event
    assert np.array_equal(dbg["xr"][use], ref["xr"]) and np.array_equal(dbg["yr"][use], ref["yr"])
    assert np.array_equal(dbg["mask"][use], ref["mask"])
    assert np.array_equal(dbg["disp"][use][ref["mask"]], ref["disp"])
    assert st.n_used == use.sum() and st.n_inliers == ref["mask"].sum() and st.n_index_errors == 0
    assert st.t_min == t[use].min() and st.t_max == t[use].max()
    # frames
    assert np.array_equal(depth == 0, ref["depth"] == 0)
    nz = ref["depth"] != 0
    rel = np.abs(depth[nz] - ref["depth"][nz]) / ref["depth"][nz]
    assert rel.max(initial=0.0) <= REL_TOL_DEPTH
    assert np.array_equal(depth, ref["depth"])  # same FP64 divide -> in fact bit-exact
    assert np.array_equal(bgr, ref["bgr"])
    # AoS entry point gives the same frame
    assert np.array_equal(depth2, depth) and np.array_equal(bgr2, bgr) and st2.n_inliers == st.n_inliers


@pytest.mark.parametrize("camera", [False, True])
def test_tiny_frame(camera):
    tb = S.make_tables(S.C_TINY)
    _check_frame(tb, S.make_events(S.C_TINY), camera)


@pytest.mark.parametrize("camera", [False, True])
def test_tiny_frame_unsorted_and_polarity(camera):
    tb = S.make_tables(S.C_TINY)
    _check_frame(tb, S.make_events(S.C_TINY, frame=3, shuffled=True, p_zero_fraction=0.05), camera, p=True)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 255, 257, 1023, 1025, 4099])
def test_ragged_sizes(n):
    tb = S.make_tables(S.C_TINY)
    _check_frame(tb, S.make_events(S.C_TINY, frame=n, n=n), False)


@pytest.mark.parametrize("camera", [False, True])
def test_c1m_full_size(camera):
    tb = S.make_tables(S.C_1M)
    _check_frame(tb, S.make_events(S.C_1M), camera)


def test_c1m_unsorted_polarity():
    tb = S.make_tables(S.C_1M)
    _check_frame(tb, S.make_events(S.C_1M, frame=1, shuffled=True, p_zero_fraction=0.05), False, p=True)


def test_empty_frame_is_defined():
    tb = S.make_tables(S.C_TINY)
    with XMapsEngine(tb) as eng:
        depth, bgr, st = eng.process_frame(np.zeros(0, np.uint16), np.zeros(0, np.uint16), np.zeros(0, np.int64))
        assert not depth.any() and (bgr == 255).all() and st.n_events == 0 and st.n_inliers == 0
        # and the engine still works afterwards
        ev = S.make_events(S.C_TINY)
        x, y, t, _ = S.to_soa(ev)
        d2, _, _ = eng.process_frame(x, y, t)
        ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
        assert np.array_equal(d2, ref["depth"])


def test_frames_back_to_back_need_no_clear():
    """The packed key carries a frame tag: consecutive frames on one handle must not leak into each other."""
    tb = S.make_tables(S.C_TINY)
    with XMapsEngine(tb) as eng:
        for f in range(6):
            ev = S.make_events(S.C_TINY, frame=f, n=500 + 700 * (f % 3))
            x, y, t, _ = S.to_soa(ev)
            d, b, _ = eng.process_frame(x, y, t)
            ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
            assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])


G1 = ["g1a_n1000", "g1b_n100000", "g1c_unsorted_dups", "g1d_rint_ties", "g1e_edges", "g1f_float32_t",
      "g1g_float64_t", "g1h_equal_t"]


@pytest.mark.parametrize("name", G1)
def test_golden_event_path(golden_dir, name):
    """HIP path vs outputs captured from the reference's own functions (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    rh, rw = int(g["rect_h"]), int(g["rect_w"])
    tb = {"cam_mapx_i16": g["mapx"], "cam_mapy_i16": g["mapy"], "proj_x_map": g["xmap"],
          "disp_proj_mapxy_i16": np.zeros((4, 4, 2), np.int16), "rect_w": rw, "rect_h": rh,
          "p03": 1.0, "z_near": 0.1, "z_far": 1.2}
    x, y, t = g["x"], g["y"], g["t"]
    with XMapsEngine(tb) as eng:
        dbg = eng.debug_event_outputs(x, y, t)
        assert np.array_equal(dbg["xr"], g["xr"]) and np.array_equal(dbg["yr"], g["yr"])
        assert np.array_equal(dbg["mask"], g["mask"])
        assert np.array_equal(dbg["disp"][g["mask"]], g["disp"])
        # stage API, reference signatures
        xr, yr = eng.rectify_cam_coords_i16(x, y)
        assert np.array_equal(xr, g["xr"]) and np.array_equal(yr, g["yr"])
        disp_full, mask = eng.event_disparity_full(xr, yr, t)
        assert np.array_equal(mask, g["mask"]) and np.array_equal(disp_full[mask], g["disp"])
        if "proj_index_error" in g.files:
            with pytest.raises(IndexError):
                eng.disp_map_projector_view(xr, yr, disp_full, mask)
        else:
            assert np.array_equal(eng.disp_map_projector_view(xr, yr, disp_full, mask), g["disp_map_proj"])
        assert np.array_equal(eng.disp_map_camera_view(x, y, disp_full, mask), g["disp_map_cam"])
    # fused camera-view frame == reference's camera-view disparity map pushed through A5
    with XMapsEngine(tb, camera_perspective=True) as eng:
        depth, _, _ = eng.process_frame(x, y, t)
        assert np.array_equal(depth, O.disparity_to_depth_rectified(g["disp_map_cam"], 1.0))


def test_golden_frame_stages(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_frame_stages.npz"))
    tb = S.make_tables(S.C_TINY)
    tb["p03"] = float(g["p03"])
    with XMapsEngine(tb) as eng:
        depth = eng.disparity_to_depth(g["disp"])
        assert np.array_equal(depth, g["depth"])
        bgr = eng.colorize_depth_from_disp(g["disp"])
    u8 = O.clip_normalize_uint8_depth_frame(g["depth"], 0.1, 1.2)
    assert np.array_equal(bgr, O.generate_color_map(u8))
    # A6/A7 against the reference's own output (golden G2), two hard assertions:
    # (1) the white mask (apply_white_mask, d2d:39-43) is where the reference's u8 frame is 0 -- exactly
    assert np.array_equal((bgr == 255).all(-1), g["u8"] == 0)
    # (2) u8 itself: the stub run multiplies `* 255` in f32 (NumPy 2), Numba types it as f64 -- at most 1 LSB apart, and
    #     only on a handful of pixels (on this fixture: none)
    du8 = np.abs(u8.astype(int) - g["u8"].astype(int))
    assert du8.max() <= 1 and (du8 != 0).sum() <= max(1, u8.size // 100)
    tb["p03"] = float(g["p03_neg"])
    with XMapsEngine(tb) as eng:
        assert np.array_equal(eng.disparity_to_depth(g["disp"]), g["depth_neg"])


def test_stage_remap_matches_oracle():
    tb = S.make_tables(S.C_TINY)
    rng = np.random.default_rng(3)
    rect = rng.integers(0, 60, (tb["rect_h"], tb["rect_w"])).astype(np.float32)
    rect[rng.random(rect.shape) < 0.8] = 0
    with XMapsEngine(tb) as eng:
        got = eng.remap_rectified_disp_map_to_proj(rect)
    assert np.array_equal(got, O.remap_rectified_disp_map_to_proj(rect, tb["disp_proj_mapxy_i16"]))


def test_out_of_sensor_event_raises_index_error():
    tb = S.make_tables(S.C_TINY)
    ev = S.make_events(S.C_TINY, n=100)
    x, y, t, _ = S.to_soa(ev)
    x = x.copy()
    x[17] = tb["cam_w"]  # one past the last column -> NumPy IndexError in rectify_cam_coords_i16
    with XMapsEngine(tb) as eng:
        with pytest.raises(IndexError):
            eng.process_frame(x, y, t)
    with pytest.raises(IndexError):
        O.rectify_cam_coords_i16(tb["cam_mapx_i16"], tb["cam_mapy_i16"], x.astype(np.int64), y.astype(np.int64))


def test_c10m_full_size_matches_c_oracle():
    """BASELINE config 4 shapes (1280x720, rect 3520x1980, 10 M events) on one GPU vs the C oracle (all host cores)."""
    from c_oracle import COracle
    cfg = S.C_10M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg)
    x, y, t, _ = S.to_soa(evs)
    ref = COracle(tb, False, omp=True).process_ev_frame(x, y, t)
    with XMapsEngine(tb) as eng:
        depth, bgr, st = eng.process_frame(x, y, t)
    assert st.n_inliers == ref["n_inliers"] and st.n_index_errors == 0
    assert np.array_equal(depth, ref["depth"]) and np.array_equal(bgr, ref["bgr"])


def test_c1m_direct_kernels_match_tiled(monkeypatch):
    """The one-thread-per-event kernels (XM_K1_DIRECT / XM_K2_DIRECT, also the automatic fallback for tables that do not
    fit LDS) give the same frame as the tiled ones."""
    tb = S.make_tables(S.C_1M)
    evs = S.make_events(S.C_1M, frame=2)
    x, y, t, _ = S.to_soa(evs)
    with XMapsEngine(tb) as eng:
        d0, b0, s0 = eng.process_frame(x, y, t)
    xm_option("XM_K1_DIRECT", "1")
    xm_option("XM_K2_DIRECT", "1")
    with XMapsEngine(tb) as eng:
        d1, b1, s1 = eng.process_frame(x, y, t)
    assert np.array_equal(d0, d1) and np.array_equal(b0, b1) and s0.n_inliers == s1.n_inliers


def test_order_invariance_property_full_size():
    """Domain property at full size: events hitting DIFFERENT cells commute.  Reversing the frame changes which event
    is 'last' per cell, but a frame whose events are de-duplicated per cell beforehand must not depend on order."""
    tb = S.make_tables(S.C_1M)
    evs = S.make_events(S.C_1M, frame=5)
    x, y, t, _ = S.to_soa(evs)
    with XMapsEngine(tb) as eng:
        dbg = eng.debug_event_outputs(x, y, t)
        m = dbg["mask"]
        cell = dbg["yr"][m].astype(np.int64) * 100000 + (dbg["xr"][m].astype(np.int64) + dbg["disp"][m])
        # keep the last event of every cell, in original order
        idx = np.nonzero(m)[0]
        _, last_pos = np.unique(cell[::-1], return_index=True)
        keep = np.sort(idx[len(idx) - 1 - last_pos])
        d_full, _, _ = eng.process_frame(x, y, t)
        # same extrema are needed for identical time columns: re-attach the frame's first and last event
        tmin_i, tmax_i = int(np.argmin(t)), int(np.argmax(t))
        keep = np.unique(np.concatenate((keep, [tmin_i, tmax_i])))
        perm = np.random.default_rng(0).permutation(len(keep))
        d_sub, _, _ = eng.process_frame(x[keep], y[keep], t[keep])
        d_perm, _, _ = eng.process_frame(x[keep][perm], y[keep][perm], t[keep][perm])
    assert np.array_equal(d_sub, d_perm)
    # and de-duplication itself does not change the frame unless the two re-attached events win a cell they lost before
    assert (d_full != d_sub).mean() < 1e-4


def test_dirty_line_flags_option_gives_the_same_frames(monkeypatch):
    """XM_K2_FLAGS=1 (K1 marks dirty key-frame lines, K2 skips clean ones): an optional byte-saving path, same frames,
    also across consecutive frames on one slot (stale flags must only ever be false positives)."""
    xm_option("XM_K2_FLAGS", "1")
    tb = S.make_tables(S.C_1M)
    with XMapsEngine(tb) as eng:
        for f, kw in ((0, {}), (1, {"shuffled": True}), (2, {"n": 50_000}), (3, {})):
            evs = S.make_events(S.C_1M, frame=f, **kw)
            x, y, t, _ = S.to_soa(evs)
            d, b, _ = eng.process_frame(x, y, t)
            ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
            assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"]), f
