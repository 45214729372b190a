"""-m gpu: the owner-tile K1 (x_maps_amd/csrc/xmaps_k1own.hpp) -- rigs on which several X-map time columns of a row share one
cell of the rectified frame (the reference's own calibration: X_MAP_WIDTH = projector_width, python/x_maps_disparity.py:58-59,
scattered through python/cam_proj_calibration.py:299-303).  A cell belongs to the tile of the FIRST column that maps to it;
tiles read a halo of the next columns' events; last-writer-wins is resolved in LDS slots indexed by the (sheared) cell; the
flush is a plain store of every owned cell.  Checked here against the CPU oracle, bit for bit: duplicates across neighbouring
columns and across tile boundaries, stale cells, x noise, unsorted streams (redo), groups, AoS, tile widths, shear on / off,
and the ESL-like rig (real calibration geometry)."""
import numpy as np
import pytest

from conftest import xm_option

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["rows", "groups"])
def _tiles_for_single_frames_too(request):
    xm_option("XM_COLS", "2")  # single-frame calls take the tiles as well (default: groups only)
    # ownership of a cell per row (2-byte flush, small halo) / per 8-row group where the rig allows it (16-byte flush; the default)
    if request.param == "rows":
        xm_option("XM_OWN_GROUPED", "0")
    return request.param


def _ref(tb, evs, **kw):
    x, y, t, _ = S.to_soa(evs)
    return O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, **kw)


def _run(eng, evs):
    x, y, t, _ = S.to_soa(evs)
    return eng.process_frame(x, y, t)


def _same(got, ref):
    d, b, st = got
    return np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"]) and st.n_inliers == int(ref["mask"].sum())


def test_shared_cell_rig_qualifies_and_matches_the_oracle():
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    with XMapsEngine(tb) as eng:
        info = eng.cols_info()
        assert info["mode"] == "own" and 3 <= info["halo"] <= 7 and info["shear_m"] != 0, info  # (3.3 columns per cell; more with ownership per 8-row group)
        for f in range(4):
            evs = S.make_events(cfg, frame=f)
            assert _same(_run(eng, evs), _ref(tb, evs)), f
        pc = eng.path_counts()
        assert pc["cols"] == 4 and pc["general"] == 0 and eng.sorted_fallbacks() == 0, pc


@pytest.mark.parametrize("w", ["4", "12", "16"])
def test_tile_widths(monkeypatch, w):
    xm_option("XM_OWN_W", w)
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    with XMapsEngine(tb) as eng:
        info = eng.cols_info()
        assert info["mode"] == "own" and info["w"] == int(w), info
        for f in range(2):
            evs = S.make_events(cfg, frame=10 + f, n=60_000)
            assert _same(_run(eng, evs), _ref(tb, evs)), f
        assert eng.path_counts()["cols"] == 2 and eng.sorted_fallbacks() == 0


def test_unsheared_frame(monkeypatch):
    """XM_OWN_SHEAR=0: the frame keeps its plain [rect_w][rect_h] layout; the bands of the (tile, 8-row group)s absorb the slant
    on their own (the shear only makes the flush's stores fall into fewer frame columns)."""
    xm_option("XM_OWN_SHEAR", "0")
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    with XMapsEngine(tb) as eng:
        info = eng.cols_info()
        assert info["mode"] == "own" and info["shear_m"] == 0 and info["shear_extra"] == 0, info
        for f in range(2):
            evs = S.make_events(cfg, frame=f)
            assert _same(_run(eng, evs), _ref(tb, evs)), f
        assert eng.path_counts()["cols"] == 2


@pytest.mark.parametrize("cpc,slant", [(2.0, 0.35), (4.6, -0.7), (1.4, 0.0), (7.5, -0.2)])
def test_other_rig_shapes(cpc, slant):
    """1.4 .. 7.5 time columns per cell (halo 1 .. 7), slanted either way or not at all."""
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg, cols_per_cell=cpc, slant=slant)
    with XMapsEngine(tb) as eng:
        info = eng.cols_info()
        assert info["mode"] == "own", info
        assert {2.0: 1, 4.6: 4, 1.4: 1, 7.5: 7}[cpc] <= info["halo"] <= 7, info  # >= the largest column distance inside a cell of a row
        for f in range(2):
            evs = S.make_events(cfg, frame=20 + f)
            assert _same(_run(eng, evs), _ref(tb, evs)), f
        assert eng.path_counts()["cols"] == 2 and eng.sorted_fallbacks() == 0


def test_more_than_eight_columns_per_cell_keeps_the_packed_keys():
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg, cols_per_cell=9.5)
    with XMapsEngine(tb) as eng:
        assert eng.cols_info()["mode"] == "none"
        evs = S.make_events(cfg, frame=1)
        assert _same(_run(eng, evs), _ref(tb, evs))
        assert eng.path_counts()["cols"] == 0


def test_no_stale_cells_when_the_content_changes_from_frame_to_frame():
    """No tag, no clear: every tile stores its empty owned cells as zeros.  Frames that cover only a part of the scan, or have a
    hole in the middle (whole tiles without an event), must not show the previous frame's cells."""
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    with XMapsEngine(tb, n_slots=1) as eng:
        for f in range(18):
            evs = S.make_events(cfg, frame=f % 3, n=40_000 + 9_000 * (f % 4))
            if f % 3 == 1:
                evs = evs[: len(evs) // 3]
            if f % 3 == 2:
                t = evs["t"].astype(np.int64)
                evs = evs[(t < t[0] + 4_000) | (t > t[0] + 9_000)]
            assert _same(_run(eng, evs), _ref(tb, evs)), f
        assert eng.sorted_fallbacks() == 0 and eng.path_counts()["cols"] == 18


def test_last_writer_across_columns_and_tile_boundaries():
    """The same camera pixel fires again one to three time columns later: both events land on the same frame cell from different
    X-map columns -- also when the two columns belong to different tiles (the later one is then a halo event of the owner)."""
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    rng = np.random.default_rng(3)
    evs = S.make_events(cfg, frame=5, n=50_000)
    n = len(evs)
    per_col = n // cfg.proj_w
    src = rng.choice(n - 4 * per_col, 6_000, replace=False)
    dst = src + rng.integers(per_col // 2, 3 * per_col, len(src))
    evs["x"][dst], evs["y"][dst] = evs["x"][src], evs["y"][src]
    # ... and some with a different x (another disparity) in the same row, so that the winner's VALUE matters
    src2 = rng.choice(n - 4 * per_col, 3_000, replace=False)
    dst2 = src2 + rng.integers(1, 2 * per_col, len(src2))
    evs["y"][dst2] = evs["y"][src2]
    evs["x"][dst2] = np.clip(evs["x"][src2].astype(np.int64) - rng.integers(0, 3, len(src2)), 0, cfg.cam_w - 1)
    with XMapsEngine(tb) as eng:
        assert _same(_run(eng, evs), _ref(tb, evs))
        assert eng.sorted_fallbacks() == 0 and eng.path_counts()["cols"] == 1


def test_x_noise_negative_disparities_and_rows_outside():
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    rng = np.random.default_rng(9)
    evs = S.make_events(cfg, frame=11, n=45_000)
    noisy = rng.random(len(evs)) < 0.05
    evs["x"][noisy] = rng.integers(0, cfg.cam_w, int(noisy.sum()))  # any disparity, many negative (xmd:29)
    with XMapsEngine(tb) as eng:
        got, ref = _run(eng, evs), _ref(tb, evs)
        assert 0 < int(ref["mask"].sum()) < len(evs)
        assert _same(got, ref)
        assert eng.sorted_fallbacks() == 0


@pytest.mark.parametrize("kind", ["swapped_blocks", "one_late_event", "reversed"])
def test_unsorted_streams_fail_the_tiles_and_are_redone_exactly(kind):
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    evs = S.make_events(cfg, frame=6, n=48_000)
    if kind == "swapped_blocks":
        a, b = evs[8_000:9_000].copy(), evs[30_000:31_000].copy()
        evs[8_000:9_000], evs[30_000:31_000] = b, a
    elif kind == "one_late_event":
        e = evs[5_000].copy()
        evs[5_000:20_000] = evs[5_001:20_001]
        evs[20_000] = e
    else:
        evs = evs[::-1].copy()
    with XMapsEngine(tb) as eng:
        assert _same(_run(eng, evs), _ref(tb, evs))
        assert eng.sorted_fallbacks() == 1


def test_groups_and_aos():
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    frames = [S.make_events(cfg, frame=40 + f, n=30_000 + 5_000 * f) for f in range(5)]
    frames[3] = frames[3][::-1].copy()  # one frame of the group is not sorted: redone
    with XMapsEngine(tb, n_slots=5) as eng:
        out = eng.process_event_frames(frames)
        for f, (d, b) in enumerate(out):
            r = _ref(tb, frames[f])
            assert np.array_equal(d, r["depth"]) and np.array_equal(b, r["bgr"]), f
        assert eng.path_counts()["cols"] == 5 and eng.sorted_fallbacks() == 1
        d, b, st = eng.process_events(frames[1])  # one AoS frame
        r = _ref(tb, frames[1])
        assert np.array_equal(d, r["depth"]) and np.array_equal(b, r["bgr"]) and st.n_inliers == int(r["mask"].sum())


def test_unaligned_soa_input_takes_the_scalar_loader():
    torch = pytest.importorskip("torch")
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    evs = S.make_events(cfg, frame=2, n=33_333)
    x, y, t, _ = S.to_soa(evs)
    dev = torch.device("cuda", 0)
    pad = 3
    X = torch.zeros(len(x) + pad, dtype=torch.int16, device=dev)
    Y = torch.zeros(len(x) + pad, dtype=torch.int16, device=dev)
    T = torch.zeros(len(x) + pad, dtype=torch.int64, device=dev)
    X[pad:] = torch.from_numpy(x.view(np.int16)).to(dev)
    Y[pad:] = torch.from_numpy(y.view(np.int16)).to(dev)
    T[pad:] = torch.from_numpy(t).to(dev)
    depth = torch.zeros((cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb) as eng:
        eng.process_frame_device(X.data_ptr() + 2 * pad, Y.data_ptr() + 2 * pad, T.data_ptr() + 8 * pad, None, len(x), depth.data_ptr())
        eng.sync()
        assert eng.path_counts()["cols"] == 1
    assert np.array_equal(depth.cpu().numpy(), _ref(tb, evs)["depth"])


def test_esl_like_rig_real_calibration_geometry():
    """BASELINE configs 1 / 3 stand-in: the reference's calibration (data/ESL_calib_hhi.yaml numbers), 1080 time columns on ~300
    frame columns, ~150 k events per frame."""
    from x_maps_amd import rig
    cp, tb, evs0, _ = rig.make_esl_like(row_stride=13)
    frames = [evs0] + [rig.render_events(cp, tb, row_stride=13, seed=s, t0_us=7_000_000 + 16_600 * s)[0] for s in (1, 2, 3)]
    with XMapsEngine(tb, n_slots=4) as eng:
        info = eng.cols_info()
        assert info["mode"] == "own" and info["halo"] in (2, 7) and info["nxs_max"] <= 24 and info["extras"] > 0, info  # (per row / per 8-row group, wide tiles)
        for f, evs in enumerate(frames[:2]):  # frame by frame
            d, b, st = eng.process_events(evs)
            r = _ref(tb, evs)
            assert np.array_equal(d, r["depth"]) and np.array_equal(b, r["bgr"]) and st.n_inliers == int(r["mask"].sum()), f
        out = eng.process_event_frames(frames)  # as a group
        for f, (d, b) in enumerate(out):
            r = _ref(tb, frames[f])
            assert np.array_equal(d, r["depth"]) and np.array_equal(b, r["bgr"]), f
        assert eng.path_counts()["cols"] == 6 and eng.sorted_fallbacks() == 0
        # a frame too dense for the wide tiles' one event pass (every fourth camera row instead of every thirteenth: ~500 k
        # events) takes the second plan's 8-column tiles, then a sparse one the first plan's again -- on the slots used above
        dense = rig.render_events(cp, tb, row_stride=4, seed=5, t0_us=9_000_000)[0]
        assert len(dense) > 350_000
        for f, evs in enumerate((dense, frames[2], dense)):
            d, b, st = eng.process_events(evs)
            r = _ref(tb, evs)
            assert np.array_equal(d, r["depth"]) and np.array_equal(b, r["bgr"]) and st.n_inliers == int(r["mask"].sum()), f
        assert eng.path_counts()["cols"] == 9 and eng.sorted_fallbacks() == 0


def _random_shared_rig(seed):
    """a shared-cell rig with a random shape, a locally perturbed X-map (cells jump, rows swap owners, undefined cells, a
    replicated border) and a random event stream: duplicates, x noise, bursts, events outside the LUT's rows"""
    rng = np.random.default_rng(9000 + seed)
    cfg = S.C_SHARED
    cpc = float(rng.choice([1.2, 1.4, 2.0, 3.3, 4.6, 6.5]))
    slant = float(rng.choice([-0.7, -0.4, -0.2, 0.0, 0.35]))
    if 18.0 + max(0.0, -slant) * cfg.rect_h + cfg.proj_w / cpc + max(0.0, slant) * cfg.rect_h >= cfg.rect_w:
        cpc = 3.3  # (the finest X-map with the steepest slant does not fit the frame)
    tb = S.make_tables_shared_cells(cfg, cols_per_cell=cpc, slant=slant)
    xm = tb["proj_x_map"].copy()
    H, W = xm.shape
    for _ in range(int(rng.integers(0, 40))):  # local jumps of a few cells (what rounding of a real time map does)
        r, c = int(rng.integers(0, H)), int(rng.integers(1, W))
        xm[r, c:c + int(rng.integers(1, 6))] += np.int16(rng.integers(-3, 4))
    if rng.random() < 0.5:  # a replicated border: the last columns all map to one far-away cell
        xm[:, W - int(rng.integers(1, 12)):] = xm[:, :1] + np.int16(rng.integers(0, 40))
    xm[rng.random(xm.shape) < 0.002] = 0  # undefined cells
    xm = np.clip(xm, 0, S.X_OFFSET + cfg.rect_w - 1).astype(np.int16)
    tb["proj_x_map"] = np.ascontiguousarray(xm)
    n = int(rng.integers(25_000, 120_000))
    evs = S.make_events(cfg, frame=int(rng.integers(0, 1000)), n=n)
    k = int(rng.integers(0, n // 50))
    if k:  # duplicates of earlier events a little later in the stream (same pixel, later stamp): last writer must win
        src = rng.integers(0, n - 1, k)
        dst = np.minimum(src + rng.integers(1, 400, k), n - 1)
        evs["x"][dst] = evs["x"][src]
        evs["y"][dst] = evs["y"][src]
    if rng.random() < 0.3:  # a burst: a tenth of the frame's events inside 1 % of its time
        i0 = int(rng.integers(0, n - n // 10))
        evs["t"][i0:i0 + n // 10] = np.sort(evs["t"][i0] + rng.integers(0, 130, n // 10))
        evs["t"] = np.maximum.accumulate(evs["t"])
    m = rng.random(n) < 0.01  # x noise far outside the window / the camera
    evs["x"][m] = rng.integers(0, cfg.cam_w, int(m.sum())).astype(np.uint16)
    return tb, evs


@pytest.mark.parametrize("seed", range(16))
def test_random_shared_cell_rigs_and_streams(seed):
    """whatever the rig and the stream: frame == oracle, through the owner tiles when the rig qualifies (single frames and a group
    of three), on the packed keys otherwise"""
    tb, evs = _random_shared_rig(seed)
    ref = _ref(tb, evs)
    with XMapsEngine(tb, n_slots=4) as eng:
        assert _same(_run(eng, evs), ref), (seed, eng.cols_info())
        if eng.path_counts()["cols"] == 1 and eng.sorted_fallbacks() == 0:  # A3's output itself, before the 7x7 maximum can hide a cell
            assert np.array_equal(eng.debug_last_disp_frame(), np.asarray(ref["disp_map"]).astype(np.uint16)), seed
        frames = [evs, evs[: len(evs) // 2].copy(), evs]
        out = eng.process_event_frames(frames)
        for e, (d, b) in zip(frames, out):
            r = _ref(tb, e)
            assert np.array_equal(d, r["depth"]) and np.array_equal(b, r["bgr"]), (seed, len(e))


@pytest.mark.parametrize("passes", [1, 2, 3, 4])
@pytest.mark.parametrize("n", [40_000, 700_000])
def test_row_passes_through_the_lds_slots(passes, n):
    """A tile's rows go through its LDS slots in 1..4 passes (round 5: a block's LDS is what limits the tiles a CU holds at once):
    sparse frames keep their events and gathers in registers between the passes, frames of more events per tile than a block
    holds (700 k events over 34 tiles of 8 columns: > 4096 each) look them up again for every row pass.  Same frames either way,
    also after each other on one slot (the flush of a pass clears its slots for the next one)."""
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    xm_option("XM_OWN_ROW_PASSES", str(passes))
    with XMapsEngine(tb, n_slots=1) as eng:
        for f in range(3):
            evs = S.make_events(cfg, frame=f, n=n)
            assert _same(_run(eng, evs), _ref(tb, evs)), (passes, n, f)
        assert eng.sorted_fallbacks() == 0 and eng.path_counts()["cols"] == 3


def test_two_plans_on_one_slot(_tiles_for_single_frames_too):
    """A rig whose slant allows ownership per 8-row group gets up to three plans: tiles of 20 and of 16 columns (16-byte flush; the
    halo grows by the slant over 8 rows) for frames whose tiles fit one event pass of a block, and the per-row plan's 8-column
    tiles for denser frames.  All rewrite every cell a pair maps to, so frames of any density may follow each other on one slot:
    every frame == oracle."""
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg, cols_per_cell=1.4, slant=0.0)
    with XMapsEngine(tb, n_slots=1) as eng:
        info = eng.cols_info()
        assert info["mode"] == "own", info
        if _tiles_for_single_frames_too == "groups":
            assert info["w"] > 8 and info["dense_w"] == 8 and info["dense_halo"] <= info["halo"], info
        else:
            assert info["w"] == 8 and info["dense_w"] == 0, info
        for f, n in enumerate([40_000, 400_000, 30_000, 60_000, 600_000, 45_000, 58_000, 500_000]):  # (<= 52 k: tiles of 20; <= 65 k: of 16; else per row)
            evs = S.make_events(cfg, frame=f, n=n)
            assert _same(_run(eng, evs), _ref(tb, evs)), (f, n)
        assert eng.sorted_fallbacks() == 0 and eng.path_counts()["cols"] == 8
