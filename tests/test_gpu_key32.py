"""-m gpu: the compact (32-bit) key frame of the verified-sorted projector-view path -- tag4 | tile | disparity, order field
= tile index (x_maps_amd/csrc/xmaps_kernels.hpp: key32_tag).  Exactness rests on three mechanisms, each exercised here:
every event of a tile is resolved in LDS (x-noise events fetch their LUT entry from global memory and join the slots),
events outside a tile's LDS time window mark the frame as failed (automatic redo on the 64-bit path), and the 4-bit tag is
kept unambiguous by clearing the frame at least every 15 frames of a slot."""
import numpy as np
import pytest

from conftest import xm_option

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _keyed_path_only(monkeypatch):
    """The column tiles (tests/test_gpu_cols.py) take precedence on rigs that qualify for both: switch them off here."""
    xm_option("XM_COLS", "0")


def _ref(tb, evs):
    x, y, t, _ = S.to_soa(evs)
    return O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)


def _run(eng, evs):
    x, y, t, _ = S.to_soa(evs)
    return eng.process_frame(x, y, t)


def test_more_than_15_frames_per_slot_and_changing_content():
    """40 consecutive frames on one slot: the 4-bit tag repeats, stale cells of 15 frames ago must never show."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    with XMapsEngine(tb) as eng:
        for f in range(40):
            evs = S.make_events(cfg, frame=f % 3, n=200_000 + 50_000 * (f % 4))
            if f % 5 == 4:  # the scan covers only part of the frame: most cells of the previous frames stay untouched
                evs = evs[: len(evs) // 3]
            d, b, st = _run(eng, evs)
            ref = _ref(tb, evs)
            assert st.n_unsorted == 0 and st.n_inliers == int(ref["mask"].sum()), f
            assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"]), f
        assert eng.sorted_fallbacks() == 0


def test_x_noise_events_join_the_lds_slots():
    """Sorted frame with events far outside the tile's 16-column LUT window (hot pixels, reflections): they fetch their LUT
    entry from global memory but are still ordered exactly against the tile's other events -- including same-cell duplicates
    with different disparities before and after them."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(9)
    evs = S.make_events(cfg, frame=11, n=400_000)
    noisy = rng.random(len(evs)) < 0.02
    evs["x"][noisy] = rng.integers(0, cfg.cam_w, int(noisy.sum()))
    # duplicates of noisy events right next to them in the stream (same pixel, same time): last writer must win
    idx = np.nonzero(noisy)[0][::7]
    idx = idx[idx + 1 < len(evs)]
    evs["x"][idx + 1], evs["y"][idx + 1] = evs["x"][idx], evs["y"][idx]
    with XMapsEngine(tb) as eng:
        d, b, st = _run(eng, evs)
        ref = _ref(tb, evs)
        assert st.n_unsorted == 0 and eng.sorted_fallbacks() == 0  # handled on the compact path, no redo
        assert st.n_inliers == int(ref["mask"].sum())
        assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])


def test_events_outside_the_time_window_force_the_exact_redo():
    """A sorted frame whose event rate collapses in the middle: the tiles there span more X-map columns than the LDS window
    holds, the frame fails the compact path and is redone on the 64-bit path -- synchronously here, asynchronously below."""
    torch = pytest.importorskip("torch")
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=12, n=300_000)
    t = evs["t"].astype(np.int64)
    mid = (t > t[0] + 5_000) & (t < t[0] + 8_000)
    keep = ~mid | (np.arange(len(evs)) % 40 == 0)  # 2.5 % of the events survive in the middle of the scan
    sparse = evs[keep]
    dense = S.make_events(cfg, frame=13, n=300_000)
    ref_s, ref_d = _ref(tb, sparse), _ref(tb, dense)
    with XMapsEngine(tb, n_slots=2) as eng:
        d, b, st = _run(eng, sparse)
        assert st.n_unsorted > 0 and eng.sorted_fallbacks() == 1
        assert np.array_equal(d, ref_s["depth"]) and np.array_equal(b, ref_s["bgr"])
        dev = torch.device("cuda", 0)
        bufs = []
        for e in (sparse, dense, sparse, dense, dense):
            x, y, tt, _ = S.to_soa(e)
            X, Y, T = (torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), tt))
            out = torch.zeros((cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
            bufs.append((X, Y, T, out))
        torch.cuda.synchronize()
        for X, Y, T, out in bufs:
            eng.process_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, len(T), out.data_ptr(), None)
        eng.sync()
        assert eng.sorted_fallbacks() == 3
        for (X, Y, T, out), r in zip(bufs, (ref_s, ref_d, ref_s, ref_d, ref_d)):
            assert np.array_equal(out.cpu().numpy(), r["depth"])


def test_compact_path_pauses_when_it_keeps_failing():
    """A stream whose every frame fails the compact path is not run twice for ever: after a few failures the handle stays
    on the 64-bit path for a while (results exact throughout)."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=14, n=300_000)
    t = evs["t"].astype(np.int64)
    mid = (t > t[0] + 5_000) & (t < t[0] + 8_000)
    sparse = evs[~mid | (np.arange(len(evs)) % 40 == 0)]
    ref = _ref(tb, sparse)
    with XMapsEngine(tb) as eng:
        for f in range(12):
            d, _, _ = _run(eng, sparse)
            assert np.array_equal(d, ref["depth"]), f
        assert 3 <= eng.sorted_fallbacks() <= 4  # then paused: the later frames ran once


def test_switch_off_gives_the_same_frames(monkeypatch):
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    evs = S.make_events(cfg, frame=15)
    with XMapsEngine(tb) as eng:
        d0, b0, s0 = _run(eng, evs)
    xm_option("XM_KEY32", "0")
    with XMapsEngine(tb) as eng:
        d1, b1, s1 = _run(eng, evs)
    assert np.array_equal(d0, d1) and np.array_equal(b0, b1) and s0.n_inliers == s1.n_inliers


def test_border_tiles_of_the_frame_kernel_on_the_compact_frame():
    """Projector maps that reach past every border of the rectified frame (tests/test_gpu_a4_bruteforce.py's tables) through
    the event path: the compact frame's cell-by-cell border loader against the oracle."""
    from test_gpu_a4_bruteforce import border_tables
    for rect_w, rect_h, proj_w, proj_h in ((176, 132, 64, 48), (200, 100, 50, 37)):
        tb, _ = border_tables(rect_w, rect_h, proj_w, proj_h, seed=rect_w)
        base = S.make_tables(S.C_TINY)
        tb["proj_x_map"] = base["proj_x_map"][:rect_h] if rect_h <= base["proj_x_map"].shape[0] else base["proj_x_map"]
        tb["rect_h"] = tb["proj_x_map"].shape[0]
        tb["cam_mapy_i16"] = np.clip(base["cam_mapy_i16"], -5, tb["rect_h"] + 5).astype(np.int16)
        evs = S.make_events(S.C_TINY, frame=2, n=40_000)
        ref = _ref(tb, evs)
        with XMapsEngine(tb) as eng:
            d, b, st = _run(eng, evs)
        assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])
