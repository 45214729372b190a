"""-m gpu: the software-pipelined K2 (x_maps_amd/csrc/xmaps_k2pipe.hpp: dilate o remap -> depth -> u8 -> Turbo BGR,
python/disp_to_depth.py:7-97) that groups of frames take on the u16 disparity frame of the column / owner tiles.  The suite's
rigs are too small for it by default (fewer than three items per persistent block keep the one-block-per-tile kernel), so
XM_K2_PIPE=2 sends every group there; xm_debug_k2_pipe_frames confirms it.  Checked against the CPU oracle, bit for bit, in
every variant of the kernel: two / four pixels per thread, strided / consecutive pixel assignment (u16 pixel table, 8 / 16-byte
depth stores, packed BGR rows), projector widths that are / are not multiples of 4, 8, 16 and of the tile width (every BGR store
width and the partial last tile), the per-disparity table in LDS cut short (larger disparities read the global table), depth only /
BGR only, output rows at odd addresses."""
import numpy as np
import pytest

from conftest import xm_option

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _ref(tb, evs):
    x, y, t, _ = S.to_soa(evs)
    return O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)


def _rig(kind, proj_w):
    if kind == "own":  # several time columns per cell: owner tiles, sheared frame
        cfg = S.RigConfig("k2p-own", 160, 128, proj_w, 120, 40_000)  # (rect_h = 352: the pipelined loader needs a multiple of 8)
        return cfg, S.make_tables_shared_cells(cfg, cols_per_cell=proj_w / 82.0)
    if kind == "fine":  # a projector finer than the rectified frame (0.7 cells per pixel, as the reference's ESL rig): small patches,
        cfg = S.RigConfig("k2p-fine", 160, 128, proj_w, 256, 60_000)  # the rigs that take 64 x 16-pixel tiles by default
        return cfg, S.make_tables_shared_cells(cfg, cols_per_cell=2.0, slant=0.0)
    if kind == "tall":  # a coarse projector: patches of 9 - 10 row octets (the loader takes the patch in memory order, <= 128 rows)
        cfg = S.RigConfig("k2p-tall", 160, 128, proj_w, 96, 60_000)
        return cfg, S.make_tables(cfg)
    cfg = S.RigConfig("k2p-cols", 160, 128, proj_w, 128, 60_000)  # 1.3 cells per time column: column tiles; patches of <= 64 rows
    return cfg, S.make_tables(cfg)


def _check_group(tb, cfg, n_frames=5, **kw):
    frames = [S.make_events(cfg, frame=70 + f, n=cfg.n_events + 3_000 * f) for f in range(n_frames)]
    frames[1] = frames[1][: len(frames[1]) // 2].copy()  # a shorter scan: stale cells must not show
    with XMapsEngine(tb, n_slots=n_frames) as eng:
        for rep in range(2):  # twice: the second group finds the first one's frames in its slots
            out = eng.process_event_frames(frames, **kw)
            for f, (d, b) in enumerate(out):
                r = _ref(tb, frames[f])
                if kw.get("want_depth", True):
                    assert np.array_equal(d, r["depth"]), (rep, f)
                if kw.get("want_bgr", True):
                    assert np.array_equal(b, r["bgr"]), (rep, f)
        assert eng.path_counts()["cols"] == 2 * n_frames and eng.sorted_fallbacks() == 0, (eng.path_counts(), eng.cols_info())
        return eng.debug_k2_pipe_frames()


@pytest.mark.parametrize("consec", ["0", "1"])
@pytest.mark.parametrize("ppt", ["2", "4"])
@pytest.mark.parametrize("kind,proj_w", [("cols", 256), ("cols", 264), ("cols", 260), ("cols", 250), ("own", 270), ("own", 320),
                                         ("fine", 640), ("fine", 600), ("fine", 604), ("fine", 570), ("tall", 256)])
def test_every_variant_against_the_oracle(monkeypatch, kind, proj_w, ppt, consec):
    xm_option("XM_K2_PIPE", "2")
    xm_option("XM_K2_PIPE_PPT", ppt)
    xm_option("XM_K2_CONSEC", consec)
    cfg, tb = _rig(kind, proj_w)
    assert _check_group(tb, cfg) == 10


def test_a_fine_projector_takes_the_wide_tiles_by_default(monkeypatch):
    xm_option("XM_K2_PIPE", "2")
    cfg, tb = _rig("fine", 640)
    assert _check_group(tb, cfg, n_frames=3) == 6


@pytest.mark.parametrize("consec", ["0", "1"])
@pytest.mark.parametrize("nlds", ["1", "24", "40"])
def test_disparities_beyond_the_lds_copy_of_the_table_read_the_global_one(monkeypatch, nlds, consec):
    xm_option("XM_K2_PIPE", "2")
    xm_option("XM_K2_NLDS_MAX", nlds)  # the shared-cell rig's disparities are around 30
    xm_option("XM_K2_CONSEC", consec)
    cfg, tb = _rig("own", 272)
    assert _check_group(tb, cfg, n_frames=3) == 6


@pytest.mark.parametrize("consec", ["0", "1"])
def test_depth_only_and_bgr_only(monkeypatch, consec):
    xm_option("XM_K2_PIPE", "2")
    xm_option("XM_K2_CONSEC", consec)
    cfg, tb = _rig("cols", 256)
    assert _check_group(tb, cfg, n_frames=3, want_bgr=False) == 6
    assert _check_group(tb, cfg, n_frames=3, want_depth=False) == 6


@pytest.mark.parametrize("consec", ["0", "1"])
@pytest.mark.parametrize("shift", [1, 4, 8])
def test_output_rows_at_any_address(monkeypatch, shift, consec):
    """the BGR rows leave as 16 / 8 / 4-byte or single-byte stores, whichever the frame's address and row length allow"""
    torch = pytest.importorskip("torch")
    xm_option("XM_K2_PIPE", "2")
    xm_option("XM_K2_CONSEC", consec)
    xm_option("XM_K2_PIPE_PPT", "4")
    cfg, tb = _rig("cols", 256)
    frames = [S.make_events(cfg, frame=90 + f) for f in range(3)]
    dev = torch.device("cuda", 0)
    n = [len(f) for f in frames]
    off = np.concatenate(([0], np.cumsum(n))).astype(np.uint64)
    cat = np.concatenate(frames)
    x, y, t, _ = S.to_soa(cat)
    X = torch.from_numpy(x.view(np.int16)).to(dev)
    Y = torch.from_numpy(y.view(np.int16)).to(dev)
    T = torch.from_numpy(t).to(dev)
    px = cfg.proj_w * cfg.proj_h
    depth = torch.zeros(3 * px, dtype=torch.float32, device=dev)
    bgr = torch.zeros(3 * px * 3 + 64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, n_slots=3) as eng:
        eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, off, depth.data_ptr(), bgr.data_ptr() + shift)
        eng.sync()
        assert eng.debug_k2_pipe_frames() == 3
    got_b = bgr.cpu().numpy()[shift: shift + 3 * px * 3].reshape(3, cfg.proj_h, cfg.proj_w, 3)
    got_d = depth.cpu().numpy().reshape(3, cfg.proj_h, cfg.proj_w)
    for f in range(3):
        r = _ref(tb, frames[f])
        assert np.array_equal(got_d[f], r["depth"]) and np.array_equal(got_b[f], r["bgr"]), f
