"""Groups of SPARSE frames (too few events per X-map time column for the tiled K1 -- the reference's own recordings look like
this: ~150 k events over 1080 columns) go through three multi-frame launches: K0 -> one-thread-per-event K1 -> K2.  Every frame
must equal the oracle's, whatever the time dtype, with and without the polarity column, SoA and AoS, both views."""
import numpy as np
import pytest

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _frames(cfg, F, rng, unsorted=()):
    evs = []
    for f in range(F):
        n = int(rng.integers(1, 9000)) if f != 2 else 1  # (frame 2: a single event)
        e = S.make_events(cfg, frame=300 + f, n=n)
        e["p"] = rng.integers(0, 2, len(e))
        if f in unsorted:
            e = e[rng.permutation(len(e))]
        evs.append(e)
    return evs


@pytest.mark.parametrize("camera", [False, True])
@pytest.mark.parametrize("t_kind", ["int64", "float32", "float64"])
@pytest.mark.parametrize("use_p", [False, True])
def test_sparse_groups_soa(camera, t_kind, use_p):
    torch = pytest.importorskip("torch")
    from x_maps_amd import _native as N
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    F = 7
    evs = _frames(cfg, F, rng, unsorted=(4,))
    H, W = (cfg.cam_h, cfg.cam_w) if camera else (cfg.proj_h, cfg.proj_w)
    t_np = {"int64": np.int64, "float32": np.float32, "float64": np.float64}[t_kind]
    t_code = {"int64": N.XM_T_INT64, "float32": N.XM_T_FLOAT32, "float64": N.XM_T_FLOAT64}[t_kind]
    lens = [len(e) for e in evs]
    lens.insert(3, 0)  # an empty frame inside the group
    offs = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    cat = np.concatenate(evs)
    x, y, t, p = S.to_soa(cat)
    t = t.astype(t_np)
    refs = []
    for e in evs:
        ex, ey, et, ep = S.to_soa(e)
        et = et.astype(t_np)
        if use_p:
            keep = ep == 1
            ex, ey, et = ex[keep], ey[keep], et[keep]
        refs.append(None if len(ex) == 0 else
                    O.process_ev_frame(tb, ex.astype(np.int64), ey.astype(np.int64), et, camera_perspective=camera))
    refs.insert(3, None)
    X = torch.from_numpy(x.view(np.int16)).to(dev)
    Y = torch.from_numpy(y.view(np.int16)).to(dev)
    T = torch.from_numpy(t).to(dev)
    P = torch.from_numpy(np.ascontiguousarray(p).astype(np.int16)).to(dev)
    depth = torch.full((F + 1, H, W), 7.0, dtype=torch.float32, device=dev)
    bgr = torch.zeros((F + 1, H, W, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, camera_perspective=camera, n_slots=F + 1) as eng:
        for rep in range(3):
            eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), P.data_ptr() if use_p else None, offs,
                                     depth.data_ptr(), bgr.data_ptr(), t_dtype=t_code)
            eng.sync()
            d, b = depth.cpu().numpy(), bgr.cpu().numpy()
            for i, r in enumerate(refs):
                if r is None:  # no event (left): an all-zero disparity frame
                    assert not d[i].any() and (b[i] == 255).all(), (rep, i)
                else:
                    assert np.array_equal(d[i], r["depth"]), (rep, i)
                    assert np.array_equal(b[i], r["bgr"]), (rep, i)
            depth.fill_(7.0)
            bgr.zero_()
            torch.cuda.synchronize()
        assert eng.path_counts()["cols"] == 0 and eng.path_counts()["key32"] == 0


@pytest.mark.parametrize("camera", [False, True])
def test_sparse_groups_aos_equal_single_frame_calls(camera):
    torch = pytest.importorskip("torch")
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    F = 6
    evs = _frames(cfg, F, rng, unsorted=(1, 5))
    H, W = (cfg.cam_h, cfg.cam_w) if camera else (cfg.proj_h, cfg.proj_w)
    offs = np.concatenate(([0], np.cumsum([len(e) for e in evs]))).astype(np.uint64)
    buf = np.empty(int(offs[-1]), S.EVENT_CD_DTYPE)
    for f, e in enumerate(evs):
        buf[int(offs[f]):int(offs[f + 1])] = e
    A = torch.from_numpy(buf.view(np.uint8).reshape(-1, 16).copy()).to(dev)
    depth = torch.zeros((F, H, W), dtype=torch.float32, device=dev)
    bgr = torch.zeros((F, H, W, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, camera_perspective=camera, n_slots=F) as eng:
        eng.process_events_batch_device(A.data_ptr(), offs, depth.data_ptr(), bgr.data_ptr())
        eng.sync()
        d, b = depth.cpu().numpy(), bgr.cpu().numpy()
        for f, e in enumerate(evs):
            x, y, t, _ = S.to_soa(e)
            r = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera)
            assert np.array_equal(d[f], r["depth"]) and np.array_equal(b[f], r["bgr"]), f
            d1, b1, st = eng.process_events(e)
            assert np.array_equal(d1, d[f]) and np.array_equal(b1, b[f]), f
            assert st.n_inliers == int(r["mask"].sum())


def test_sparse_groups_inside_a_hipgraph():
    """The same three launches captured: a graph of sparse frames replays bit-exactly, frame statistics included."""
    torch = pytest.importorskip("torch")
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(13)
    F = 5
    evs = _frames(cfg, F, rng)
    offs = np.concatenate(([0], np.cumsum([len(e) for e in evs]))).astype(np.uint64)
    x, y, t, _ = S.to_soa(np.concatenate(evs))
    X = torch.from_numpy(x.view(np.int16)).to(dev)
    Y = torch.from_numpy(y.view(np.int16)).to(dev)
    T = torch.from_numpy(t).to(dev)
    depth = torch.zeros((F, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    bgr = torch.zeros((F, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, n_slots=F) as eng:
        g = eng.graph_create(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(), bgr.data_ptr())
        for rep in range(3):
            g.launch()
            eng.sync()
            torch.cuda.synchronize()
            d, b = depth.cpu().numpy(), bgr.cpu().numpy()
            for f, e in enumerate(evs):
                ex, ey, et, _ = S.to_soa(e)
                r = O.process_ev_frame(tb, ex.astype(np.int64), ey.astype(np.int64), et)
                assert np.array_equal(d[f], r["depth"]) and np.array_equal(b[f], r["bgr"]), (rep, f)
            depth.zero_()
            torch.cuda.synchronize()
        g.close()


@pytest.mark.parametrize("camera", [False, True])
@pytest.mark.parametrize("aos", [False, True])
def test_sparse_frames_take_the_verified_shortcut_and_unsorted_ones_are_redone(camera, aos):
    """Sparse frames skip K0 as well: the one-thread-per-event K1 takes (t[0], t[n-1]) and verifies every event against them; a
    frame that is not sorted (its extrema sit in the middle) is detected and redone with K0, automatically, exactly once."""
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(21)
    frames = []
    for f in range(8):
        e = S.make_events(cfg, frame=500 + f, n=int(rng.integers(2, 6000)))
        if f in (2, 5):
            e = e[rng.permutation(len(e))]
        if f == 6:  # only the LAST stamp is too small: every other event lies inside [t[0], t[n-1]]... except that t[n-1] < t[0]
            e["t"][-1] = e["t"][0] - 5
        frames.append(e)
    with XMapsEngine(tb, camera_perspective=camera, n_slots=3) as eng:
        for rep in range(2):
            for i, e in enumerate(frames):
                if aos:
                    d, b, st = eng.process_events(e)
                else:
                    x, y, t, _ = S.to_soa(e)
                    d, b, st = eng.process_frame(x, y, t)
                x, y, t, _ = S.to_soa(e)
                r = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera)
                assert np.array_equal(d, r["depth"]) and np.array_equal(b, r["bgr"]), (rep, i)
                assert st.n_inliers == int(r["mask"].sum()) and st.t_min == float(t.min()) and st.t_max == float(t.max()), (rep, i)
        assert eng.sorted_fallbacks() == 2 * 3
        pc = eng.path_counts()
        assert pc["sorted_key64"] >= 2 * 8 and pc["key32"] == 0 and pc["cols"] == 0
