"""-m gpu: XM_FLAG_ADAPTIVE_BATCH -- asynchronous one-frame-per-call submissions (xm_process_frame on device pointers, what an
offline replay issues back to back) are grouped into multi-frame launches whenever the GPU is still busy with earlier frames, and
launched at once when it is not.  Whatever the grouping turns out to be, every frame's outputs must equal the oracle's after
xm_sync(); frames that fail the sorted-order verification are redone; synchronous calls in between flush what is held back."""
import numpy as np
import pytest

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ref(tb, evs, **kw):
    x, y, t, _ = S.to_soa(evs)
    return O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, **kw)


def _upload(evs, dev):
    x, y, t, _ = S.to_soa(evs)
    return (torch.from_numpy(x.view(np.int16)).to(dev), torch.from_numpy(y.view(np.int16)).to(dev), torch.from_numpy(t).to(dev))


@pytest.mark.parametrize("camera", [False, True])
def test_frames_submitted_back_to_back_equal_the_oracle(camera):
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    F = 40
    host = [S.make_events(cfg, frame=f % 5, n=300_000 + 50_000 * (f % 3)) for f in range(F)]
    host[7] = host[7][::-1].copy()  # one frame is not sorted: its tiles object, it is redone on the general path
    refs = {f: _ref(tb, host[f], camera_perspective=camera) for f in (0, 1, 7, 8, 23, F - 1)}
    up = [_upload(e, dev) for e in host]
    H, W = (cfg.cam_h, cfg.cam_w) if camera else (cfg.proj_h, cfg.proj_w)
    depth = torch.zeros((F, H, W), dtype=torch.float32, device=dev)
    bgr = torch.zeros((F, H, W, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, camera_perspective=camera, n_slots=F, adaptive_batch=True) as eng:
        for rep in range(2):
            for f in range(F):
                x, y, t = up[f]
                eng.process_frame_device(x.data_ptr(), y.data_ptr(), t.data_ptr(), None, len(host[f]), depth[f].data_ptr(), bgr[f].data_ptr())
            eng.sync()
            for f, r in refs.items():
                assert np.array_equal(depth[f].cpu().numpy(), r["depth"]), (rep, f)
                assert np.array_equal(bgr[f].cpu().numpy(), r["bgr"]), (rep, f)
            depth.zero_()
            bgr.zero_()
            torch.cuda.synchronize()
        assert eng.sorted_fallbacks() == 2  # frame 7, once per repetition
        pc = eng.path_counts()
        assert sum(pc.values()) >= 2 * F  # every frame went through some K1 (redone frames count twice)


def test_a_synchronous_call_flushes_what_is_held_back_and_few_slots_switch_it_off():
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    frames = [S.make_events(cfg, frame=f, n=20_000) for f in range(12)]
    up = [_upload(e, dev) for e in frames]
    depth = torch.zeros((12, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for slots in (8, 4):  # (4 slots: the flag is ignored, every frame is launched at once)
        with XMapsEngine(tb, n_slots=slots, adaptive_batch=True) as eng:
            for f in range(12):
                x, y, t = up[f]
                eng.process_frame_device(x.data_ptr(), y.data_ptr(), t.data_ptr(), None, len(frames[f]), depth[f].data_ptr(), None)
                if f == 5:  # a synchronous host-memory frame in the middle: everything before it has been submitted when it returns
                    d, b, st = eng.process_events(frames[0])
                    assert np.array_equal(d, _ref(tb, frames[0])["depth"])
            st = eng.last_frame_stats()  # (flushes, waits for the last frame)
            assert st.n_inliers == int(_ref(tb, frames[11])["mask"].sum())
            eng.sync()
            for f in (0, 4, 5, 6, 11):
                assert np.array_equal(depth[f].cpu().numpy(), _ref(tb, frames[f])["depth"]), (slots, f)
        depth.zero_()
        torch.cuda.synchronize()


def test_eventcd_records_on_an_owner_tile_rig():
    """xm_process_frame_aos on a handle with the flag, on a rig whose X-map is not injective (owner tiles): frames of different
    lengths, one of them unsorted, submitted back to back; every frame == the oracle (what `bench.py --esl` times as
    other_modes.one_frame_per_call)."""
    cfg = S.C_SHARED
    tb = S.make_tables_shared_cells(cfg)
    dev = torch.device("cuda", 0)
    F = 24
    host = [S.make_events(cfg, frame=f % 6, n=40_000 + 7_000 * (f % 4)) for f in range(F)]
    host[5] = host[5][::-1].copy()
    up = [torch.from_numpy(e.view(np.uint8).reshape(-1, 16).copy()).to(dev) for e in host]
    depth = torch.zeros((F, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    bgr = torch.zeros((F, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, n_slots=F, adaptive_batch=True) as eng:
        assert eng.cols_info()["mode"] == "own"
        for f in range(F):
            eng.process_events_device(up[f].data_ptr(), len(host[f]), False, depth[f].data_ptr(), bgr[f].data_ptr())
        eng.sync()
        for f in (0, 4, 5, 6, 13, F - 1):
            r = _ref(tb, host[f])
            assert np.array_equal(depth[f].cpu().numpy(), r["depth"]), f
            assert np.array_equal(bgr[f].cpu().numpy(), r["bgr"]), f
        assert eng.sorted_fallbacks() == 1
        assert eng.path_counts()["cols"] >= F - 1


def test_frames_of_different_layouts_never_share_a_group():
    """SoA int64, AoS records, SoA float64 time stamps and SoA with a polarity column submitted in turn on one adaptive handle:
    a group's kernels are instantiated for ONE layout (chosen from its first frame), so a frame of another layout must close the
    pending group first -- every frame still equals the oracle"""
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    F = 24
    host = [S.make_events(cfg, frame=f % 4, n=200_000 + 30_000 * (f % 3), p_zero_fraction=0.2 if f % 4 == 3 else 0.0) for f in range(F)]
    keep = []
    depth = torch.zeros((F, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, n_slots=32, adaptive_batch=True) as eng:
        for rep in range(2):
            for f in range(F):
                e = host[f]
                x, y, t, p = S.to_soa(e)
                kind = f % 4
                if kind == 0:  # SoA, int64 time stamps
                    bufs = (torch.from_numpy(x.view(np.int16)).to(dev), torch.from_numpy(y.view(np.int16)).to(dev), torch.from_numpy(t).to(dev))
                    eng.process_frame_device(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), None, len(e), depth[f].data_ptr(), None)
                elif kind == 1:  # 16-byte EventCD records
                    bufs = (torch.from_numpy(e.view(np.uint8).reshape(-1, 16).copy()).to(dev),)
                    eng.process_events_device(bufs[0].data_ptr(), len(e), False, depth[f].data_ptr(), None)
                elif kind == 2:  # SoA, float64 time stamps (the evaluation caller's dtype)
                    bufs = (torch.from_numpy(x.view(np.int16)).to(dev), torch.from_numpy(y.view(np.int16)).to(dev),
                            torch.from_numpy(t.astype(np.float64)).to(dev))
                    from x_maps_amd import _native as N
                    eng.process_frame_device(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), None, len(e), depth[f].data_ptr(), None,
                                             t_dtype=N.XM_T_FLOAT64)
                else:  # SoA with a polarity column (20 % of the events have p = 0)
                    bufs = (torch.from_numpy(x.view(np.int16)).to(dev), torch.from_numpy(y.view(np.int16)).to(dev), torch.from_numpy(t).to(dev),
                            torch.from_numpy(p).to(dev))
                    eng.process_frame_device(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), len(e),
                                             depth[f].data_ptr(), None)
                keep.append(bufs)  # (inputs stay untouched until sync)
            eng.sync()
            for f in range(F):
                e = host[f]
                x, y, t, p = S.to_soa(e)
                tt = t.astype(np.float64) if f % 4 == 2 else t
                if f % 4 == 3:  # the polarity column: events[p == 1], as the pipe's PolarityFilterAlgorithm(1) leaves them
                    x, y, tt = x[p == 1], y[p == 1], tt[p == 1]
                ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), tt, want_bgr=False)
                assert np.array_equal(depth[f].cpu().numpy(), ref["depth"]), (rep, f, f % 4)
            depth.zero_()
            torch.cuda.synchronize()
            keep.clear()
