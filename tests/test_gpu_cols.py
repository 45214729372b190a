"""-m gpu: the column-tile K1 (x_maps_amd/csrc/xmaps_k1cols.hpp) -- a tile = W X-map time columns and the index range of the
sorted stream that falls into them (found by a search over t), last-writer-wins resolved entirely in the tile's LDS slots,
the flush a plain store of every live slot into a plain u16 disparity frame that K2 reads untagged.  Exactness rests on:
the injectivity check of xm_create, every event being verified against its tile's columns (failures -> automatic redo on
the general path), empty slots being stored as zeros (no stale cells from earlier frames), and the search being a
deterministic function of (stream, column).  Each is exercised here against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from conftest import xm_option

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _column_tiles_for_single_frames_too(monkeypatch):
    """By default only groups of frames (xm_process_batch) take the column tiles -- a single frame's boundary pass is a third
    dependent launch; XM_COLS=2 sends single-frame calls there as well, which is how most cases below reach the kernel."""
    xm_option("XM_COLS", "2")


def _ref(tb, evs, **kw):
    x, y, t, _ = S.to_soa(evs)
    return O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, **kw)


def _run(eng, evs):
    x, y, t, _ = S.to_soa(evs)
    return eng.process_frame(x, y, t)


def _same(got, ref):
    d, b, st = got
    return np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"]) and st.n_inliers == int(ref["mask"].sum())


def test_dense_frames_take_the_column_tiles_and_match_the_oracle():
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    with XMapsEngine(tb) as eng:
        for f in range(3):
            evs = S.make_events(cfg, frame=f)
            assert _same(_run(eng, evs), _ref(tb, evs)), f
        pc = eng.path_counts()
        assert pc["cols"] == 3 and pc["key32"] == 0 and pc["general"] == 0 and eng.sorted_fallbacks() == 0


def test_no_stale_cells_when_the_content_changes_from_frame_to_frame():
    """Consecutive frames of very different coverage on ONE slot: there is no tag and no clear, every tile stores its empty
    slots as zeros -- a frame whose scan covers only part of the time axis must not show the previous frame's cells."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    with XMapsEngine(tb, n_slots=1) as eng:
        for f in range(24):
            evs = S.make_events(cfg, frame=f % 3, n=500_000 + 100_000 * (f % 4))
            if f % 3 == 1:  # events only in the first third of the scan's time range (t[0], t[n-1] shrink with it)
                evs = evs[: len(evs) // 3]
            if f % 3 == 2:  # a hole in the middle of the scan: whole tiles without a single event
                t = evs["t"].astype(np.int64)
                evs = evs[(t < t[0] + 4_000) | (t > t[0] + 9_000)]
            assert _same(_run(eng, evs), _ref(tb, evs)), f
        assert eng.sorted_fallbacks() == 0 and eng.path_counts()["cols"] >= 20  # (the sparsest frames go to the direct kernel)


def test_uneven_event_rate_defeats_the_interpolated_guess_not_the_result():
    """80 % of the events in the first fifth of the scan: the interpolation window of the boundary search misses, the 64-ary
    search over the whole stream finds the same boundaries; overfull tiles take more than one pass."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(5)
    n = 900_000
    evs = S.make_events(cfg, frame=3, n=n)
    t_rel = np.sort(np.concatenate([rng.integers(0, 2_600, int(n * 0.8)), rng.integers(2_600, 13_000, n - int(n * 0.8))]))
    evs["t"] = 5_000_000 + t_rel
    evs["x"] = np.clip(np.rint(t_rel / 13_000 * cfg.cam_w + rng.normal(0.0, 2.0, n)), 0, cfg.cam_w - 1).astype(np.uint16)
    with XMapsEngine(tb) as eng:
        assert _same(_run(eng, evs), _ref(tb, evs))
        assert eng.sorted_fallbacks() == 0 and eng.path_counts()["cols"] == 1


def test_x_noise_and_duplicates_are_ordered_exactly_inside_the_tile():
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(9)
    evs = S.make_events(cfg, frame=11, n=700_000)
    noisy = rng.random(len(evs)) < 0.02
    evs["x"][noisy] = rng.integers(0, cfg.cam_w, int(noisy.sum()))
    idx = np.nonzero(noisy)[0][::7]
    idx = idx[idx + 1 < len(evs)]
    evs["x"][idx + 1], evs["y"][idx + 1] = evs["x"][idx], evs["y"][idx]  # same pixel right behind a noisy event
    with XMapsEngine(tb) as eng:
        assert _same(_run(eng, evs), _ref(tb, evs))
        assert eng.sorted_fallbacks() == 0 and eng.path_counts()["cols"] == 1


def test_ties_across_tile_boundaries():
    """Coarse time stamps: thousands of events share one stamp, stamps sit exactly on column boundaries (rint ties)."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=4, n=800_000)
    t = evs["t"].astype(np.int64)
    evs["t"] = t[0] + (t - t[0]) // 16 * 16
    with XMapsEngine(tb) as eng:
        assert _same(_run(eng, evs), _ref(tb, evs))
        assert eng.sorted_fallbacks() == 0


@pytest.mark.parametrize("kind", ["swapped_blocks", "one_late_event", "reversed"])
def test_unsorted_streams_fail_the_tiles_and_are_redone_exactly(kind):
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=6, n=600_000)
    if kind == "swapped_blocks":  # two far-apart blocks of the stream exchanged; t[0] and t[n-1] still the extrema
        a, b = evs[100_000:110_000].copy(), evs[400_000:410_000].copy()
        evs[100_000:110_000], evs[400_000:410_000] = b, a
    elif kind == "one_late_event":  # a single event delivered 3 ms late
        e = evs[50_000].copy()
        evs[50_000:200_000] = evs[50_001:200_001]
        evs[200_000] = e
    else:
        evs = evs[::-1].copy()
    ref = _ref(tb, evs)
    with XMapsEngine(tb) as eng:
        d, b, st = _run(eng, evs)
        assert st.n_unsorted > 0 and eng.sorted_fallbacks() == 1
        assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"]) and st.n_inliers == int(ref["mask"].sum())


def test_a_burst_in_one_time_column_overflows_the_tile_and_is_redone():
    """100 k events with one time stamp: more than a tile's 16-bit local index holds -> the tile objects, the frame is redone."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=7, n=500_000)
    evs["t"][200_000:300_000] = evs["t"][200_000]
    ref = _ref(tb, evs)
    with XMapsEngine(tb) as eng:
        d, b, st = _run(eng, evs)
        assert eng.sorted_fallbacks() == 1
        assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])


def test_asynchronous_frames_with_failures_in_between():
    torch = pytest.importorskip("torch")
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    good = [S.make_events(cfg, frame=20 + i, n=400_000 + 100_000 * i) for i in range(3)]
    bad = S.make_events(cfg, frame=30, n=500_000)
    a, b = bad[10_000:20_000].copy(), bad[300_000:310_000].copy()
    bad[10_000:20_000], bad[300_000:310_000] = b, a
    seq = [good[0], bad, good[1], good[2], bad, good[0], good[1]]
    refs = [_ref(tb, e, want_bgr=False)["depth"] for e in seq]
    dev = torch.device("cuda", 0)
    with XMapsEngine(tb, n_slots=3) as eng:
        bufs = []
        for e in seq:
            x, y, t, _ = S.to_soa(e)
            X, Y, T = (torch.from_numpy(v).to(dev) for v in (x.view(np.int16), y.view(np.int16), t))
            bufs.append((X, Y, T, torch.zeros((cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)))
        torch.cuda.synchronize()
        for X, Y, T, out in bufs:
            eng.process_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, len(T), out.data_ptr(), None)
        eng.sync()
        assert eng.sorted_fallbacks() == 2
        for i, ((_, _, _, out), r) in enumerate(zip(bufs, refs)):
            assert np.array_equal(out.cpu().numpy(), r), i


def test_groups_of_frames_in_one_launch():
    torch = pytest.importorskip("torch")
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    B = 4
    frames = [S.make_events(cfg, frame=40 + i, n=600_000) for i in range(B)]
    n = 600_000
    dev = torch.device("cuda", 0)
    X = torch.empty(B * n, dtype=torch.int16, device=dev)
    Y = torch.empty_like(X)
    T = torch.empty(B * n, dtype=torch.int64, device=dev)
    for i, e in enumerate(frames):
        x, y, t, _ = S.to_soa(e)
        X[i * n:(i + 1) * n] = torch.from_numpy(x.view(np.int16))
        Y[i * n:(i + 1) * n] = torch.from_numpy(y.view(np.int16))
        T[i * n:(i + 1) * n] = torch.from_numpy(t)
    depth = torch.zeros((B, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    bgr = torch.zeros((B, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
    offs = np.arange(B + 1, dtype=np.uint64) * n
    with XMapsEngine(tb, n_slots=2 * B) as eng:
        for rep in range(3):
            eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(), bgr.data_ptr())
        eng.sync()
        assert eng.path_counts()["cols"] == 3 * B and eng.sorted_fallbacks() == 0
    for i, e in enumerate(frames):
        ref = _ref(tb, e)
        assert np.array_equal(depth[i].cpu().numpy(), ref["depth"]) and np.array_equal(bgr[i].cpu().numpy(), ref["bgr"]), i


def test_eventcd_records_and_unaligned_columns():
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=8, n=700_001)
    ref = _ref(tb, evs)
    with XMapsEngine(tb) as eng:
        d, b, st = eng.process_events(evs)  # 16-byte AoS records
        assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])
        x, y, t, _ = S.to_soa(evs)
        xo, yo = np.empty(len(x) + 1, np.uint16)[1:], np.empty(len(y) + 3, np.uint16)[3:]  # 2-byte aligned only
        xo[:], yo[:] = x, y
        d2, b2, _ = eng.process_frame(xo, yo, t)
        assert np.array_equal(d2, ref["depth"]) and np.array_equal(b2, ref["bgr"])
        assert eng.path_counts()["cols"] == 2 and eng.sorted_fallbacks() == 0


def test_a_non_injective_x_map_keeps_the_keyed_paths():
    """Two time columns of a row that map to the same frame cell: xm_create must notice and leave the plain column tiles off --
    the owner tiles take such a rig where their tables fit (since round 5's row passes through the LDS slots: this one), else
    the keyed paths."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    xm = tb["proj_x_map"].copy()
    xm[400:900, 301] = xm[400:900, 300]
    tb["proj_x_map"] = xm
    evs = S.make_events(cfg, frame=9, n=600_000)
    with XMapsEngine(tb) as eng:
        assert _same(_run(eng, evs), _ref(tb, evs))
        pc = eng.path_counts()
        mode = eng.cols_info()["mode"]
        assert mode in ("own", "none"), mode
        if mode == "own":
            assert pc["cols"] == 1 and pc["key32"] == 0 and eng.sorted_fallbacks() == 0
        else:  # neither compact path: the tile-ordered 32-bit keys need the same property
            assert pc["cols"] == 0 and pc["key32"] == 0 and pc["sorted_key64"] == 1


def test_switches(monkeypatch):
    """XM_COLS=2: every qualifying frame; default: groups only (single frames keep the compact key frame); 0: off."""
    torch = pytest.importorskip("torch")
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=15)
    x, y, t, _ = S.to_soa(evs)
    dev = torch.device("cuda", 0)
    X, Y, T = (torch.from_numpy(v).to(dev) for v in (np.tile(x.view(np.int16), 2), np.tile(y.view(np.int16), 2), np.tile(t, 2)))
    out = torch.zeros((2, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    offs = np.array([0, len(t), 2 * len(t)], dtype=np.uint64)
    res = {}
    for mode in ("2", None, "0"):
        if mode is None:
            xm_option("XM_COLS", None)
        else:
            xm_option("XM_COLS", mode)
        with XMapsEngine(tb, n_slots=2) as eng:
            d, b, st = _run(eng, evs)
            single = eng.path_counts()
            eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, out.data_ptr(), None)
            eng.sync()
            both = eng.path_counts()
            res[mode] = (d, b, st.n_inliers, single, both, out.cpu().numpy().copy())
    assert res["2"][3]["cols"] == 1 and res["2"][4]["cols"] == 3
    assert res[None][3]["cols"] == 0 and res[None][3]["key32"] == 1 and res[None][4]["cols"] == 2
    assert res["0"][4]["cols"] == 0 and res["0"][4]["key32"] == 3
    for mode in (None, "0"):
        assert np.array_equal(res["2"][0], res[mode][0]) and np.array_equal(res["2"][1], res[mode][1])
        assert res["2"][2] == res[mode][2] and np.array_equal(res["2"][5], res[mode][5])
    assert np.array_equal(res["2"][5][0], res["2"][0]) and np.array_equal(res["2"][5][1], res["2"][0])


def test_index_errors_are_counted_like_the_reference():
    """Events outside the camera and events whose frame column leaves the rectified frame: dropped and counted."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=10, n=600_000)
    evs["x"][1000::50_000] = cfg.cam_w + 3
    evs["y"][2000::60_000] = cfg.cam_h
    x, y, t, _ = S.to_soa(evs)
    with XMapsEngine(tb) as eng:
        d, b, st = eng.process_frame(x, y, t, raise_on_index_error=False)
        assert st.n_index_errors > 0 and eng.path_counts()["cols"] == 1
        ok = (x < cfg.cam_w) & (y < cfg.cam_h)
        ref = _ref(tb, evs[ok])  # the reference raises; without the offending events it yields the same frame
        # (t[0] and t[n-1] are unchanged by the removal)
        assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])


def test_c10m_tiles_of_one_column_and_several_passes():
    """C-10M: 7 800 events per time column -> W = 1, two passes per tile; against the C oracle."""
    from c_oracle import COracle
    cfg = S.C_10M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=0)
    x, y, t, _ = S.to_soa(evs)
    ref = COracle(tb, False, omp=True).process_ev_frame(x, y, t, want_events=False)
    with XMapsEngine(tb) as eng:
        d, b, st = eng.process_frame(x, y, t)
        assert eng.path_counts()["cols"] == 1 and eng.sorted_fallbacks() == 0
    assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])


@pytest.mark.parametrize("n_slots", [3, 4, 8])
def test_hipgraph_batches_redo_failed_frames_on_the_device(n_slots):
    """Inside a captured batch no host is at hand to redo a frame: the graph carries the column tiles AND the 64-bit path, and
    the kernels decide per frame on the device (frame_attempt_failed).  Frames whose tiles object -- blocks of the stream
    swapped, a burst in one time column -- come out exact next to frames that take the tiles, replay after replay."""
    torch = pytest.importorskip("torch")
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    n = 600_000
    frames = [S.make_events(cfg, frame=50 + i, n=n) for i in range(6)]
    a, b = frames[1][100_000:110_000].copy(), frames[1][400_000:410_000].copy()
    frames[1][100_000:110_000], frames[1][400_000:410_000] = b, a  # not sorted
    frames[4]["t"][200_000:300_000] = frames[4]["t"][200_000]       # 100 k events in one time column
    F = len(frames)
    dev = torch.device("cuda", 0)
    X = torch.empty(F * n, dtype=torch.int16, device=dev)
    Y = torch.empty_like(X)
    T = torch.empty(F * n, dtype=torch.int64, device=dev)
    refs = []
    for i, e in enumerate(frames):
        x, y, t, _ = S.to_soa(e)
        X[i * n:(i + 1) * n] = torch.from_numpy(x.view(np.int16))
        Y[i * n:(i + 1) * n] = torch.from_numpy(y.view(np.int16))
        T[i * n:(i + 1) * n] = torch.from_numpy(t)
        refs.append(_ref(tb, e))
    depth = torch.zeros((F, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    bgr = torch.zeros((F, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    offs = np.arange(F + 1, dtype=np.uint64) * n
    with XMapsEngine(tb, n_slots=n_slots, default_priority_streams=True) as eng:
        g = eng.graph_create(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(), bgr.data_ptr())
        # groups of >= 2 frames are captured with the tiles in front; 3 slots = groups of one frame: K0 -> K1 -> K2 (64-bit keys)
        assert eng.path_counts()["cols"] == (F if n_slots >= 4 else 0)
        for rep in range(3):
            g.launch()
            eng.sync()
            d, bb = depth.cpu().numpy(), bgr.cpu().numpy()
            for f in range(F):
                assert np.array_equal(d[f], refs[f]["depth"]), (rep, f)
                assert np.array_equal(bb[f], refs[f]["bgr"]), (rep, f)
            depth.zero_()
            bgr.zero_()
            torch.cuda.synchronize()
        g.close()
        # eager groups and single frames on the same handle afterwards (host-side redo again)
        eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs[:3], depth.data_ptr(), None)
        eng.sync()
        for f in range(2):
            assert np.array_equal(depth[f].cpu().numpy(), refs[f]["depth"]), f
        assert eng.sorted_fallbacks() == 1
        x, y, t, _ = S.to_soa(frames[5])
        d5, _, st5 = eng.process_frame(x, y, t)
        assert np.array_equal(d5, refs[5]["depth"]) and st5.n_inliers == int(refs[5]["mask"].sum())


def test_live_cells_outside_the_frame_are_index_errors():
    """An X-map whose entries in a band of rows point past the right edge of the rectified frame: xm_create sees live pairs
    without a cell, the kernel keeps its per-event cell test (the other rigs of this file skip it), and an event that lands
    there is dropped and counted -- NumPy's IndexError in `frame[rows, cols] = values` (cam_proj_calibration.py:299-303)."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    xm = tb["proj_x_map"].copy()
    band = slice(300, 320)
    xm[band, 1:] = np.int16(S.X_OFFSET + cfg.rect_w + 7)  # frame column rect_w + 7: outside (and not a legal negative wrap)
    tb["proj_x_map"] = xm
    evs = S.make_events(cfg, frame=12, n=600_000)
    x, y, t, _ = S.to_soa(evs)
    xr, yr = O.rectify_cam_coords_i16(tb["cam_mapx_i16"], tb["cam_mapy_i16"], x.astype(np.int64), y.astype(np.int64))
    offending = (yr >= band.start) & (yr < band.stop)  # every such event has disp = rect_w + 7 - xr >= 0: an inlier without a cell
    offending[0] = offending[-1] = False                # (keep t[0], t[n-1])
    keep = ~offending
    ts = O.time_to_xmap_column(t, tb["t_px_scale"])
    n_err = int((offending & (ts >= 1)).sum())          # column 0 of the X-map is undefined (0): disp < 0 there, no write
    with XMapsEngine(tb) as eng:
        d, b, st = eng.process_frame(x, y, t, raise_on_index_error=False)
        assert eng.path_counts()["cols"] == 1 and eng.sorted_fallbacks() == 0
    kept = evs[keep]
    # the kept stream has the same first / last stamp, hence the same columns; the offending events are simply absent
    ref = _ref(tb, kept)
    assert st.n_index_errors == n_err and n_err > 1000
    assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])


@pytest.mark.parametrize("dims", [(80, 60, 80, 60), (96, 50, 96, 50), (128, 64, 100, 70), (64, 48, 64, 48), (200, 37, 150, 41)])
def test_small_and_odd_rigs(dims):
    """Rectified heights that are odd / not a multiple of 4 or 8 (K2's 16-byte, 8-byte and cell-by-cell loaders of the u16
    frame), X-maps shorter than a block has threads, tiles of up to 16 columns, AoS and SoA, several frames per slot."""
    cw, ch, pw, ph = dims
    rng = np.random.default_rng(cw * 1000 + ch)
    cfg = S.RigConfig(f"t{cw}x{ch}", cw, ch, pw, ph, 0)
    tb = S.make_tables(cfg)
    with XMapsEngine(tb, n_slots=2) as eng:
        for f in range(5):
            n = int(rng.integers(30_000, 90_000))
            evs = S.make_events(cfg, frame=f, n=n)
            ref = _ref(tb, evs)
            if f % 2:
                d, b, st = eng.process_events(evs)
            else:
                d, b, st = _run(eng, evs)
            assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"]) and st.n_inliers == int(ref["mask"].sum()), f
        pc = eng.path_counts()
        assert pc["cols"] + pc["key32"] + pc["sorted_key64"] + pc["general"] == 5 + eng.sorted_fallbacks()
        assert pc["cols"] >= 4, pc  # these rigs qualify (affine X-map: injective); a frame may still fail a tile and be redone


def test_offline_replay_of_event_frames_through_groups(monkeypatch):
    """DepthReprojectionPipe.process_ev_frames: a list of EventCD frames (as the trigger finder cuts them) through groups of
    multi-frame launches, one callback per frame, the same frames as process_ev_frame gives one by one -- an unsorted frame in
    the middle included (redone on the general path)."""
    xm_option("XM_COLS", None)  # library defaults: groups take the column tiles
    from x_maps_amd.depth_reprojection_pipe import DepthReprojectionPipe
    from x_maps_amd.depth_reprojection_processor import RuntimeParams
    from x_maps_amd.stats import StatsPrinter
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    frames = [S.make_events(cfg, frame=60 + i, n=300_000 + 40_000 * i) for i in range(7)]
    a, b = frames[3][10_000:20_000].copy(), frames[3][200_000:210_000].copy()
    frames[3][10_000:20_000], frames[3][200_000:210_000] = b, a
    params = RuntimeParams(camera_width=cfg.cam_w, camera_height=cfg.cam_h, projector_width=cfg.proj_w, projector_height=cfg.proj_h,
                           projector_fps=60, z_near=0.1, z_far=1.2, calib=None, projector_time_map=None, no_frame_dropping=True,
                           camera_perspective=False, tables=tb)
    got = []
    pipe = DepthReprojectionPipe(params, StatsPrinter(), got.append)
    pipe.replay_group = 4  # 7 frames: a group of 4 and one of 3
    try:
        pipe.process_ev_frames(frames)
        assert len(got) == len(frames)
        for f, (evs, bgr) in enumerate(zip(frames, got)):
            assert np.array_equal(bgr, _ref(tb, evs)["bgr"]), f
        pc = pipe._replay_engine.path_counts()
        assert pc["cols"] == 7 and pipe._replay_engine.sorted_fallbacks() == 1
        one = []
        pipe.frame_callback = one.append
        pipe.process_ev_frame(frames[5])
        assert np.array_equal(one[0], got[5])
    finally:
        pipe.close()


def _random_tile_case(seed):
    """Random rig with an injective X-map (slope > 1 frame column per time column), LUT entries outside the frame, undefined
    X-map cells, a projector map that partly points outside; a dense stream: sorted or not, with ties, x noise, duplicates."""
    rng = np.random.default_rng(7000 + seed)
    cam_w, cam_h = int(rng.integers(24, 120)), int(rng.integers(16, 90))
    rect_w, rect_h = int(rng.integers(2 * cam_w, 3 * cam_w + 8)), int(rng.integers(cam_h + 2, 3 * cam_h + 8))
    proj_w, proj_h = int(rng.integers(8, 90)), int(rng.integers(8, 70))
    xmap_w = int(rng.integers(8, 64))
    ys, xs = np.mgrid[0:cam_h, 0:cam_w]
    sx, sy = rect_w / cam_w, rect_h / cam_h
    mapx = np.rint(rng.uniform(0.5, 0.9) * sx * xs + rng.uniform(0, 6) + rng.uniform(-0.05, 0.05) * ys).astype(np.int16)
    mapy = np.rint(rng.uniform(0.7, 1.1) * sy * ys + rng.uniform(-5, 3) + rng.uniform(-0.1, 0.1) * xs).astype(np.int16)
    xmap_h = max(3, rect_h + int(rng.integers(-3, 4)))
    yr, tc = np.mgrid[0:xmap_h, 0:xmap_w]
    slope = rng.uniform(1.05, 0.9 * rect_w / xmap_w)  # > 1: two time columns of a row never share a frame column
    xmap = np.rint(4242 + rng.uniform(0, 0.08) * rect_w + tc * slope + rng.uniform(-0.05, 0.05) * yr).astype(np.int16)
    xmap[rng.random(xmap.shape) < rng.uniform(0, 0.2)] = 0
    if rng.random() < 0.5:
        xmap[:, 0] = 0
    vs, us = np.mgrid[0:proj_h, 0:proj_w]
    pm = np.stack((np.rint(us * rect_w / proj_w * rng.uniform(0.8, 1.2) + rng.uniform(-5, 5)),
                   np.rint(vs * rect_h / proj_h * rng.uniform(0.8, 1.2) + rng.uniform(-5, 5))), -1).astype(np.int16)
    tb = {"cam_w": cam_w, "cam_h": cam_h, "proj_w": proj_w, "proj_h": proj_h, "rect_w": rect_w, "rect_h": rect_h,
          "cam_mapx_i16": mapx, "cam_mapy_i16": mapy, "proj_x_map": np.ascontiguousarray(xmap),
          "disp_proj_mapxy_i16": np.ascontiguousarray(pm), "t_px_scale": xmap_w - 1, "x_offset": 4242,
          "p03": float(rng.uniform(5, 300)), "z_near": 0.1, "z_far": float(rng.uniform(0.5, 3.0))}
    n = int(rng.integers(1100 * xmap_w // 2, 2500 * xmap_w))
    span = int(rng.integers(50, 30_000))
    t_rel = np.sort(rng.integers(0, span, n))
    kind = rng.random()
    if kind < 0.15:
        t_rel = rng.permutation(t_rel)
    elif kind < 0.3:  # a few late events
        idx = rng.integers(0, n, 5)
        t_rel[idx] = t_rel[np.minimum(idx + n // 3, n - 1)]
    x = np.clip(np.rint(t_rel / span * cam_w + rng.normal(0, rng.uniform(0.5, 6), n)), 0, cam_w - 1)
    evs = np.zeros(n, S.EVENT_CD_DTYPE)
    evs["x"], evs["y"] = x, rng.integers(0, cam_h, n)
    evs["t"] = int(rng.integers(0, 2 ** 40)) + t_rel
    evs["p"] = 1
    return tb, evs


@pytest.mark.parametrize("seed", range(40))
def test_random_rigs_and_streams_on_the_tiles(seed):
    tb, evs = _random_tile_case(seed)
    x, y, t, _ = S.to_soa(evs)
    try:
        ref, ref_err = _ref(tb, evs), None
    except IndexError:
        ref, ref_err = None, IndexError
    with XMapsEngine(tb, n_slots=2) as eng:
        if ref_err is IndexError:
            with pytest.raises(IndexError):
                eng.process_frame(x, y, t)
            return
        for aos in (False, True):
            d, b, st = eng.process_events(evs) if aos else eng.process_frame(x, y, t)
            assert st.n_inliers == int(ref["mask"].sum()), aos
            assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"]), aos
            if eng.path_counts()["cols"] == 1 + aos and eng.sorted_fallbacks() == 0:  # A3's output itself, before the 7x7 maximum
                assert np.array_equal(eng.debug_last_disp_frame(), np.asarray(ref["disp_map"]).astype(np.uint16)), aos
        pc = eng.path_counts()
        # sorted streams on these rigs take the tiles; a failing frame is redone on the general path (still exact)
        assert pc["cols"] + pc["key32"] + pc["sorted_key64"] + pc["general"] == 2 + eng.sorted_fallbacks()


@pytest.mark.parametrize("xmap_w", [2, 7, 64, 640, 1080])
def test_integer_thresholds_reproduce_the_float64_column_of_every_time_stamp(xmap_w):
    """K1 never converts a time stamp: it compares a = t - tmin with thr[c].  For spans small enough to enumerate, thr[] must
    reproduce NumPy's column of EVERY integer stamp in [tmin, tmax] (rint ties included); larger spans are checked at the
    thresholds themselves and their neighbours."""
    cfg = S.RigConfig("thr", 64, 48, xmap_w, 48, 0)
    tb = S.make_tables(cfg)
    assert tb["proj_x_map"].shape[1] == xmap_w
    rng = np.random.default_rng(xmap_w)
    with XMapsEngine(tb) as eng:
        spans = [0, 1, 2, xmap_w - 1, 2 * (xmap_w - 1), 3 * (xmap_w - 1) + 1, 12_999, 16_600, 99_991] + \
                [int(v) for v in rng.integers(1, 200_000, 6)]
        for span in spans:
            for t0 in (0, 5_000_000, int(rng.integers(0, 2 ** 40)), -12_345):
                thr = eng.debug_cols_thresholds(t0, t0 + span)
                a = np.arange(span + 1, dtype=np.int64)
                col = O.time_to_xmap_column(t0 + a, xmap_w - 1).astype(np.int64)  # tmin, tmax = the array's first / last element
                want = np.searchsorted(col, np.arange(xmap_w + 1), side="left")    # first a with column >= c; span + 1 if none
                assert np.array_equal(thr, want.astype(np.uint32)), (span, t0)
        for span in (2 ** 31 + 12_345, 4_000_000_000, 999_999_937):  # too long to enumerate: check around every threshold
            t0 = 7_777
            thr = eng.debug_cols_thresholds(t0, t0 + span).astype(np.int64)
            probe = np.unique(np.clip(np.concatenate([thr - 1, thr, thr + 1, [0, span]]), 0, span))
            col = O.time_to_xmap_column(np.concatenate([[t0], t0 + probe, [t0 + span]]), xmap_w - 1).astype(np.int64)[1:-1]
            for c in range(xmap_w + 1):
                below, at = probe < thr[c], probe >= thr[c]
                assert (col[below] < c).all() and (col[at] >= c).all(), (span, c)
