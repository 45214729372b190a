"""-m gpu: the compact key frame of the CAMERA view -- (event index + 1) << 12 | disparity, one u32 per camera pixel, no tag
(x_maps_amd/csrc/xmaps_kernels.hpp: KEY32_DISP_BITS).  The order field is the event itself, so last-writer-wins holds for every
pair of writers of a pixel whatever tile they are in and wherever they sit in the stream (strays included); stale pixels cannot
show because the frame kernel zeroes every pixel it reads.  Each is exercised here against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from conftest import xm_option

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _ref(tb, evs):
    x, y, t, _ = S.to_soa(evs)
    return O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=True)


def _check(eng, tb, evs, aos=False, label=None):
    if aos:
        d, b, st = eng.process_events(evs)
    else:
        x, y, t, _ = S.to_soa(evs)
        d, b, st = eng.process_frame(x, y, t)
    ref = _ref(tb, evs)
    assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"]), label
    assert st.n_inliers == int(ref["mask"].sum()), label


def test_dense_camera_frames_take_the_compact_frame_and_leave_no_stale_pixels():
    """One slot, frames of changing extent: a scan that covers a third of the camera after a full one must show empty pixels where
    the previous frame had events (the frame kernel zeroed them), duplicates on a pixel resolve to the last event."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(3)
    with XMapsEngine(tb, camera_perspective=True) as eng:
        for f in range(12):
            evs = S.make_events(cfg, frame=f % 3, n=660_000 + 100_000 * (f % 3))
            if f % 4 == 1:
                evs = evs[: len(evs) // 3]  # the scan stops early: two thirds of the camera columns stay dark
            if f % 4 == 2:  # many events on few pixels, in both orders of disparity
                hot = rng.random(len(evs)) < 0.05
                evs["x"][hot] = evs["x"][hot] // 8 * 8
                evs["y"][hot] = evs["y"][hot] // 8 * 8
            _check(eng, tb, evs, aos=bool(f & 1), label=f)
        assert eng.sorted_fallbacks() == 0
        pc = eng.path_counts()
        assert pc["key32"] == 12 and pc["general"] == 0 and pc["sorted_key64"] == 0


def test_strays_and_unsorted_frames():
    """x noise far outside the tile's camera-column window goes through the global path of the same kernel with the same key;
    a shuffled frame fails the (t[0], t[n-1]) verification and is redone on the 64-bit path -- and the frames after it are right."""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(4)
    with XMapsEngine(tb, camera_perspective=True, n_slots=2) as eng:
        evs = S.make_events(cfg, frame=21, n=500_000)
        noisy = rng.random(len(evs)) < 0.03
        evs["x"][noisy] = rng.integers(0, cfg.cam_w, int(noisy.sum()))
        idx = np.nonzero(noisy)[0][::5]
        idx = idx[idx + 1 < len(evs)]
        evs["x"][idx + 1], evs["y"][idx + 1] = evs["x"][idx], evs["y"][idx]  # same pixel right behind a stray: last writer wins
        _check(eng, tb, evs, label="noise")
        shuffled = S.make_events(cfg, frame=22, n=400_000)
        shuffled = shuffled[rng.permutation(len(shuffled))]
        _check(eng, tb, shuffled, label="shuffled")
        _check(eng, tb, shuffled, aos=True, label="shuffled aos")
        for f in range(4):
            _check(eng, tb, S.make_events(cfg, frame=30 + f, n=350_000), label=("after", f))
        assert eng.sorted_fallbacks() == 2


def test_groups_of_camera_frames_and_the_event_index_limit():
    """xm_process_batch in the camera view (multi-frame K1 / K2 on the compact frames); a frame of 2^20 events or more cannot
    use a 20-bit order field and takes the 64-bit frame, with the same result."""
    torch = pytest.importorskip("torch")
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    frames = [S.make_events(cfg, frame=40 + f, n=260_000 + 40_000 * f) for f in range(4)]
    offs = np.concatenate(([0], np.cumsum([len(e) for e in frames]))).astype(np.uint64)
    x, y, t, _ = S.to_soa(np.concatenate(frames))
    X = torch.from_numpy(x.view(np.int16)).to(dev)
    Y = torch.from_numpy(y.view(np.int16)).to(dev)
    T = torch.from_numpy(t).to(dev)
    depth = torch.zeros((4, cfg.cam_h, cfg.cam_w), dtype=torch.float32, device=dev)
    bgr = torch.zeros((4, cfg.cam_h, cfg.cam_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, camera_perspective=True, n_slots=4) as eng:
        for rep in range(3):
            eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(), bgr.data_ptr())
            eng.sync()
            for f, e in enumerate(frames):
                ref = _ref(tb, e)
                assert np.array_equal(depth[f].cpu().numpy(), ref["depth"]) and np.array_equal(bgr[f].cpu().numpy(), ref["bgr"]), (rep, f)
            depth.zero_()
            torch.cuda.synchronize()
        assert eng.path_counts()["key32"] == 12
        big = S.make_events(cfg, frame=50, n=(1 << 20) + 5)
        _check(eng, tb, big, label="2^20 + 5 events")
        assert eng.path_counts()["key32"] == 12 and eng.path_counts()["sorted_key64"] == 1
        _check(eng, tb, frames[1], label="compact again")
        assert eng.path_counts()["key32"] == 13


def test_switch(monkeypatch):
    xm_option("XM_KEY32", "0")
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    with XMapsEngine(tb, camera_perspective=True) as eng:
        _check(eng, tb, S.make_events(cfg, frame=60, n=300_000))
        assert eng.path_counts()["key32"] == 0 and eng.path_counts()["sorted_key64"] == 1
