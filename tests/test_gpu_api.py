"""-m gpu: the reference-shaped API (pipe / processor / stage classes), the async + hipGraph paths and the
shard entry points, all through the C-ABI, against the CPU oracle."""
import numpy as np
import pytest

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S
from x_maps_amd.depth_reprojection_pipe import DepthReprojectionPipe
from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor, RuntimeParams
from x_maps_amd.stats import StatsPrinter

pytestmark = pytest.mark.gpu


def _params(cfg, tables, camera=False):
    return RuntimeParams(camera_width=cfg.cam_w, camera_height=cfg.cam_h, projector_width=cfg.proj_w,
                         projector_height=cfg.proj_h, projector_fps=60, z_near=0.1, z_far=1.2, calib=None,
                         projector_time_map=None, no_frame_dropping=True, camera_perspective=camera, tables=tables)


def _ref(tb, evs, camera=False):
    x, y, t, _ = S.to_soa(evs)
    return O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera)


@pytest.mark.parametrize("camera", [False, True])
@pytest.mark.parametrize("fused", [True, False])
def test_pipe_process_ev_frame_calls_back_with_reference_frame(camera, fused):
    tb = S.make_tables(S.C_TINY)
    evs = S.make_events(S.C_TINY)
    got = []
    pipe = DepthReprojectionPipe(_params(S.C_TINY, tb, camera), StatsPrinter(), got.append)
    pipe.fused = fused
    pipe.process_ev_frame(evs)
    ref = _ref(tb, evs, camera)
    assert len(got) == 1 and got[0].dtype == np.uint8 and got[0].shape == ref["bgr"].shape
    assert np.array_equal(got[0], ref["bgr"])
    assert np.array_equal(pipe.depth_frame(evs), ref["depth"])
    with pytest.raises(ValueError):  # t.min() of an empty frame, as in the reference
        pipe.process_ev_frame(evs[:0])
    pipe.close()


def test_stage_classes_have_reference_signatures_and_results():
    tb = S.make_tables(S.C_TINY)
    evs = S.make_events(S.C_TINY, frame=4)
    pipe = DepthReprojectionPipe(_params(S.C_TINY, tb), StatsPrinter(), lambda f: None)
    ref = _ref(tb, evs)
    xr, yr = pipe.calib_maps.rectify_cam_coords_i16(evs)
    disp, mask = pipe.x_maps_disp.compute_event_disparity(events=evs, ev_x_rect_i16=xr, ev_y_rect_i16=yr)
    assert disp.dtype == np.int16 and mask.dtype == bool
    assert np.array_equal(xr, ref["xr"]) and np.array_equal(disp, ref["disp"]) and np.array_equal(mask, ref["mask"])
    dm = pipe.calib_maps.compute_disp_map_projector_view(xr, yr, mask, disp)
    assert np.array_equal(dm, ref["disp_map"])
    pd = pipe.disp_to_depth.remap_rectified_disp_map_to_proj(dm)
    assert np.array_equal(pd, ref["proj_disp"])
    assert np.array_equal(pipe.disp_to_depth.colorize_depth_from_disp(pd), ref["bgr"])
    dc = pipe.calib_maps.compute_disp_map_camera_view(evs, mask, disp)
    assert np.array_equal(dc, _ref(tb, evs, True)["disp_map"])
    pipe.close()


@pytest.mark.parametrize("activity", [True, False])
def test_processor_context_manager_end_to_end_stream(activity):
    """Packets -> polarity filter -> activity filter (the default, as in the reference; one GPU call per packet on this path)
    -> trigger finder -> hot path -> window.show_async, like the reference's loop (pipe:110-119) -- the HOST chain
    (RuntimeParams(device_ingest=False), the opt-out since round 6: the default path is tests/test_gpu_ingest.py's)."""
    import ingest_oracle as IO
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(7)
    chunks = []
    for f in range(6):
        n = 9000 if activity else 2500
        start = 1_000_000 + f * 16_600
        tt = np.unique(np.concatenate((np.sort(rng.integers(0, 13_000, n)) + start, np.arange(start, start + 13_000, 25))))
        ev = np.zeros(len(tt), S.EVENT_CD_DTYPE)
        ev["t"] = tt
        ev["x"] = np.clip((tt - start) / 13_000 * cfg.cam_w + rng.normal(0, 1.5, len(tt)), 0, cfg.cam_w - 1).astype(np.uint16)
        ev["y"] = rng.integers(0, cfg.cam_h, len(tt))
        ev["p"] = (rng.random(len(tt)) < 0.9)
        chunks.append(ev)
    stream = np.concatenate(chunks)
    frames_seen = []

    params = _params(cfg, tb)
    params.device_ingest = False
    if not activity:
        params.activity_filter = False
    with DepthReprojectionProcessor(params) as proc:
        assert (proc._pipe.activity_filter is not None) == activity and proc._pipe.ingest is None
        orig = proc._pipe.process_ev_frame

        def spy(evs):
            frames_seen.append(evs.copy())
            orig(evs)

        proc._pipe.trigger_finder.frame_callback = spy
        packet = int(1e6 / 60 / 4)
        edges = np.arange(stream["t"][0], stream["t"][-1] + packet, packet)
        cuts = np.searchsorted(stream["t"], edges)
        for a, b in zip(cuts[:-1], cuts[1:]):
            proc.process_events(stream[a:b])
            assert not proc.should_close()
        assert proc.stats_printer.counters["frames shown"] == len(frames_seen) >= 1  # (the finder loses lock easily: by design)
        assert proc.stats_printer.counters["processed evs"] == len(stream)
        last = proc._window.last_frame
    assert (frames_seen[-1]["p"] == 1).all()
    assert np.array_equal(last, _ref(tb, frames_seen[-1])["bgr"])
    # the same frames as the CPU chain cuts from the same packets
    tf = IO.TriggerFinderOracle(60)
    act = IO.ActivityFilterC(cfg.cam_w, cfg.cam_h, int(1e6 / 60))
    for a, b in zip(cuts[:-1], cuts[1:]):
        pos = IO.polarity_filter(stream[a:b])
        tf.process_events(act.process(pos) if activity else pos)
    same = lambda a, b: len(a) == len(b) and all(np.array_equal(a[k], b[k]) for k in ("x", "y", "t", "p"))  # (record layouts differ: padding)
    assert len(tf.frames) == len(frames_seen) and all(same(a, b) for a, b in zip(tf.frames, frames_seen))
    if activity:
        assert sum(len(f) for f in frames_seen) < (stream["p"] == 1).sum() * 0.99  # (the filter did drop isolated events)


def test_device_async_slots_and_graph_replay():
    """Device-resident frames through the async path (4 slots) and through a captured hipGraph, twice."""
    torch = pytest.importorskip("torch")
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    F, n = 6, 3000
    evs = [S.make_events(cfg, frame=f, n=n) for f in range(F)]
    refs = [_ref(tb, e) for e in evs]
    cols = [S.to_soa(e) for e in evs]
    X = torch.from_numpy(np.concatenate([c[0] for c in cols]).view(np.int16)).to(dev)
    Y = torch.from_numpy(np.concatenate([c[1] for c in cols]).view(np.int16)).to(dev)
    T = torch.from_numpy(np.concatenate([c[2] for c in cols])).to(dev)
    depth = torch.zeros((F, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    bgr = torch.zeros((F, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with XMapsEngine(tb, n_slots=4) as eng:
        for rep in range(2):
            for f in range(F):
                eng.process_frame_device(X[f * n:].data_ptr(), Y[f * n:].data_ptr(), T[f * n:].data_ptr(), None, n,
                                         depth[f].data_ptr(), bgr[f].data_ptr())
            eng.sync()
            for f in range(F):
                assert np.array_equal(depth[f].cpu().numpy(), refs[f]["depth"]) and np.array_equal(bgr[f].cpu().numpy(), refs[f]["bgr"])
            depth.zero_()
            bgr.zero_()
            torch.cuda.synchronize()
        g = eng.graph_create(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, np.arange(F + 1) * n, depth.data_ptr(),
                             bgr.data_ptr())
        for rep in range(3):
            g.launch()
            eng.sync()
            for f in range(F):
                assert np.array_equal(depth[f].cpu().numpy(), refs[f]["depth"]), (rep, f)
                assert np.array_equal(bgr[f].cpu().numpy(), refs[f]["bgr"]), (rep, f)
            depth.zero_()
            torch.cuda.synchronize()
        # eager frames still correct after graph replays advanced the device-side tags
        eng.process_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, n, depth[0].data_ptr(), None)
        eng.sync()
        assert np.array_equal(depth[0].cpu().numpy(), refs[0]["depth"])
        st = eng.profile_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, n, depth[1].data_ptr(), None)
        # (no K0: the verified (t[0], t[n-1]) shortcut holds for sparse frames too, the one-thread-per-event K1 checks it)
        assert st.n_inliers == int(refs[0]["mask"].sum()) and st.gpu_ms[0] == 0.0 and all(ms > 0 for ms in st.gpu_ms[1:])
        g.close()


@pytest.mark.parametrize("camera", [False, True])
def test_shard_calls_merge_to_the_single_frame(camera):
    """Two index shards scattered into private key frames, max-merged (what the RCCL all-reduce does), finished."""
    torch = pytest.importorskip("torch")
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    evs = S.make_events(cfg, frame=9, n=3001)
    ref = _ref(tb, evs, camera)
    x, y, t, _ = S.to_soa(evs)
    from x_maps_amd.sharded import GpuShardProvider, shard_bounds
    with XMapsEngine(tb, camera_perspective=camera) as eng:
        prov = GpuShardProvider(eng, dev)
        kfs, mms = [], []
        for r in range(2):
            a, b = shard_bounds(len(t), r, 2)
            sh = tuple(torch.from_numpy(c[a:b].copy().view(np.int16) if c.dtype == np.uint16 else c[a:b].copy()).to(dev)
                       for c in (x, y, t)) + (None,)
            kfs.append((sh, a, prov.new_key_frame()))
            mm_r = prov.new_minmax_buffer(sh)
            prov.minmax_into(sh, mm_r)  # {tmin, -tmax} left in device memory, no host round trip
            mms.append(mm_r)
        empty = (None, None, torch.zeros(0, dtype=torch.int64, device=dev), None)
        mm_e = prov.new_minmax_buffer(empty)
        prov.minmax_into(empty, mm_e)
        # the host-pointer variant (xm_shard_minmax) still exists and agrees
        mm_host = eng.shard_minmax(kfs[0][0][2].data_ptr(), None, len(kfs[0][0][2]))
        eng.sync()
        assert mm_e.cpu().tolist() == [np.iinfo(np.int64).max] * 2  # neutral for the MIN all-reduce
        assert mm_host[0] == mms[0][0].item() and mm_host[1] == -mms[0][1].item()
        with prov.collective_stream():
            mm = torch.minimum(mms[0], mms[1])  # what the MIN all-reduce of the 16-byte buffers yields
        eng.sync()
        assert mm[0].item() == t.min() and -mm[1].item() == t.max()
        for tag in (1, 2):  # second round reuses the key frames without clearing them
            for sh, a, kf in kfs:
                prov.scatter(sh, a, mm, tag, kf)
            eng.sync()
            with prov.collective_stream():  # as ShardedFrameProcessor does: the merge is ordered on the engine's stream
                merged = torch.maximum(kfs[0][2], kfs[1][2])
            depth, bgr = prov.finish(merged, tag)
            eng.sync()
            assert np.array_equal(depth.cpu().numpy(), ref["depth"]) and np.array_equal(bgr.cpu().numpy(), ref["bgr"])
            kf_ref = O.key_frame(tb, x.astype(np.int64), y.astype(np.int64), t, t.min(), t.max(), tag=tag,
                                 camera_perspective=camera)
            # the projector-view key frame lives column-major in HBM ([col][row]); the camera-view one row-major
            assert np.array_equal(merged.cpu().numpy().astype(np.uint64), kf_ref if camera else kf_ref.T)


def test_time_sorted_mode_is_exact_and_verified():
    """XM_FLAG_TIME_SORTED: sorted frames skip the extrema pass and give the same frame; an unsorted frame is detected on
    the device -- redone transparently by the synchronous call, reported by xm_sync for asynchronous ones."""
    torch = pytest.importorskip("torch")
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    # >= 18.7 K events so that the tiled kernel (which carries the verified shortcut) is chosen for this 64-column X-map
    srt = S.make_events(cfg, frame=1, n=24_000)
    uns = S.make_events(cfg, frame=2, n=24_000, shuffled=True)
    with XMapsEngine(tb, assume_time_sorted=True, n_slots=2) as eng:
        for evs, expect_flag in ((srt, False), (uns, True), (srt, False)):
            x, y, t, _ = S.to_soa(evs)
            ref = _ref(tb, evs)
            d, b, st = eng.process_frame(x, y, t)
            assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])
            assert (st.n_unsorted > 0) == expect_flag and st.n_inliers == int(ref["mask"].sum())
            assert st.t_min == t.min() and st.t_max == t.max() and st.n_used == len(t)
            d2, _, st2 = eng.process_events(evs)
            assert np.array_equal(d2, ref["depth"]) and (st2.n_unsorted > 0) == expect_flag
        eng.sync()  # synchronous fallbacks already handled: no error pending
        # asynchronous path: the sorted frame is fine, the unsorted one makes xm_sync fail loudly
        dev = torch.device("cuda", 0)
        out = torch.zeros((cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
        for evs, bad in ((srt, False), (uns, True)):
            x, y, t, _ = S.to_soa(evs)
            X, Y, T = (torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t))
            torch.cuda.synchronize()
            eng.process_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, len(t), out.data_ptr(), None)
            if bad:
                with pytest.raises(ValueError):
                    eng.sync()
                eng.sync()  # the error is reported once
            else:
                eng.sync()
                assert np.array_equal(out.cpu().numpy(), _ref(tb, evs)["depth"])


@pytest.mark.parametrize("sorted_mode", [False, True])
def test_frame_tag_wraparound_clears_the_key_frame(sorted_mode):
    """The packed key carries a 19-bit frame tag; after 2^19 - 1 frames on a slot the host enqueues one clear and the
    tags restart.  Run a slot through the wrap (asynchronous tiny frames) and check frames on both sides of it."""
    torch = pytest.importorskip("torch")
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    frames = []
    n_ev = 20_000 if sorted_mode else 300  # sorted mode: enough events for the tiled kernel (tag derived from tag_b)
    for f in range(3):
        evs = S.make_events(cfg, frame=20 + f, n=n_ev)
        x, y, t, _ = S.to_soa(evs)
        frames.append((tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)), _ref(tb, evs)["depth"]))
    out = torch.zeros((cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    total = (1 << 19) + 40
    with XMapsEngine(tb, n_slots=1, assume_time_sorted=sorted_mode) as eng:
        for i in range(total):
            (X, Y, T), ref = frames[i % 3]
            eng.process_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, n_ev, out.data_ptr(), None)
            if i in (0, 1, (1 << 19) - 3, (1 << 19) - 2, (1 << 19) - 1, 1 << 19, (1 << 19) + 1, total - 1) or i % 100_000 == 0:
                eng.sync()
                assert np.array_equal(out.cpu().numpy(), ref), i
        eng.sync()


def test_frame_event_filters_match_reference_golden(golden_dir):
    """N3: the four de-duplication filters on the GPU vs outputs of the reference's classes (golden G5)."""
    import os
    from x_maps_amd import frame_event_filter as F
    g = np.load(os.path.join(golden_dir, "g5_filters.npz"))
    ev = np.zeros(len(g["t"]), S.EVENT_CD_DTYPE)
    ev["x"], ev["y"], ev["t"], ev["p"] = g["x"], g["y"], g["t"], g["p"]
    tb = S.make_tables(S.C_TINY)
    with XMapsEngine(tb) as eng:
        for cls in ("LastEventPerXYFilter", "FirstEventPerXYFilter", "MeanFirstLastEventPerXYFilter", "FirstEventPerYTFilter"):
            out = getattr(F, cls)(eng).filter_events(ev, g["xp"])
            assert out.dtype == S.EVENT_CD_DTYPE
            for fld in ("x", "y", "t", "p"):
                assert np.array_equal(out[fld], g[f"{cls}_{fld}"]), (cls, fld)
        # the semantics the class names promise (NOT what the reference computes, see x_maps_amd/frame_event_filter.py)
        pos = ev[ev["p"] == 1]
        first = np.full((pos["y"].max() + 1, pos["x"].max() + 1), -1, np.int64)
        for i in range(len(pos) - 1, -1, -1):
            first[pos["y"][i], pos["x"][i]] = pos["t"][i]
        out = F.FirstEventPerXYFilter(eng, intended_semantics=True).filter_events(ev, None)
        assert np.array_equal(out["t"], first[first >= 0])
        last = np.full_like(first, -1)
        last[pos["y"], pos["x"]] = pos["t"]
        out = F.MeanFirstLastEventPerXYFilter(eng, intended_semantics=True).filter_events(ev, None)
        assert np.array_equal(out["t"], (first[first >= 0] + last[last >= 0]) // 2)
        # timestamps beyond 2^31 go through the reference's int32 maps: wrap, like NumPy's unsafe setitem cast
        big = ev.copy()
        big["t"] += (1 << 31) + 12345
        out = F.LastEventPerXYFilter(eng).filter_events(big, None)
        pos = big[big["p"] == 1]
        m = np.zeros((pos["y"].max() + 1, pos["x"].max() + 1), np.int32)
        m[pos["y"], pos["x"]] = pos["t"].astype(np.int32)
        k = np.zeros_like(m, bool)
        k[pos["y"], pos["x"]] = True
        assert np.array_equal(out["t"], m[k].astype(np.int64))


def test_pipe_cycles_frame_event_filters_like_the_reference():
    tb = S.make_tables(S.C_TINY)
    evs = S.make_events(S.C_TINY, frame=6)
    got = []
    st = StatsPrinter()
    pipe = DepthReprojectionPipe(_params(S.C_TINY, tb), st, got.append)
    names = []
    for _ in range(5):
        pipe.process_ev_frame(evs)
        pipe.select_next_frame_event_filter()
        names.append(st.logs[-1])
    assert [n.split(": ")[1] for n in names] == ["FirstEventPerYTFilter", "FirstEventPerXYFilter", "LastEventPerXYFilter",
                                                 "MeanFirstLastEventPerXYFilter", "NoFilter"]
    assert len(got) == 5 and np.array_equal(got[0], _ref(tb, evs)["bgr"])
    # LastEventPerXY: equivalent to running the oracle on the de-duplicated, raster-ordered events
    pos = evs[evs["p"] == 1]
    m = np.full((pos["y"].max() + 1, pos["x"].max() + 1), -1, np.int64)
    m[pos["y"], pos["x"]] = pos["t"]
    yy, xx = np.nonzero(m >= 0)
    dedup = np.zeros(len(yy), S.EVENT_CD_DTYPE)
    dedup["x"], dedup["y"], dedup["t"], dedup["p"] = xx, yy, m[yy, xx], 1
    assert np.array_equal(got[3], _ref(tb, dedup)["bgr"])
    pipe.close()


def test_sharded_processor_with_real_nccl_group(tmp_path):
    """x_maps_amd/sharded.py end to end on the GPU with a real RCCL process group (world_size 1 here; the gloo test covers
    world_size 2 logic): ExternalStream ordering of the collectives, device tensors, GpuShardProvider."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from x_maps_amd.sharded import GpuShardProvider, ShardedFrameProcessor
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        with XMapsEngine(tb) as eng:
            proc = ShardedFrameProcessor(GpuShardProvider(eng, dev), dist)
            for f in range(3):
                evs = S.make_events(cfg, frame=30 + f, n=2000 + 500 * f, shuffled=(f == 1))
                x, y, t, _ = S.to_soa(evs)
                sh = tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)) + (None,)
                torch.cuda.synchronize()
                depth, bgr = proc.process_shard(sh, 0)
                eng.sync()
                torch.cuda.synchronize()
                ref = _ref(tb, evs)
                assert np.array_equal(depth.cpu().numpy(), ref["depth"]) and np.array_equal(bgr.cpu().numpy(), ref["bgr"])
    finally:
        if created:
            dist.destroy_process_group()


def test_esl_like_rig_parity_and_accuracy():
    """BASELINE configs 1/3 stand-in: the reference's real calibration geometry (tests/golden/g6_esl_calib.npz), tables
    from the cv2-free builder (X-map built on the GPU), events rendered from a 3-D scene with microsecond stamps.
    HIP path == oracle bit-exact, and the depth it reports is the scene's depth (integer-disparity quantisation)."""
    from x_maps_amd import rig
    cp, tb, evs, gt = rig.make_esl_like(row_stride=13)
    assert 120_000 < len(evs) < 200_000 and (np.diff(evs["t"]) >= 0).all()
    ref = _ref(tb, evs)
    # the X-map the GPU built equals the oracle's construction from the same rectified time map
    xm, _ = O.compute_x_map_from_time_map(tb["time_map_rect"], tb["x_map_width"], tb["t_px_scale"], 4242, cp.projector_width)
    assert np.array_equal(xm, tb["proj_x_map"])
    with XMapsEngine(tb) as eng:
        depth, bgr, st = eng.process_events(evs)
    # (with the rectification pinned to OpenCV's -- tests/test_calibration_cpu.py -- a fifth of this rig's camera image is
    #  rectified to rows outside the 1760 x 1320 frame the reference's launch configuration allots: those events fail xmd:23)
    assert st.n_inliers == int(ref["mask"].sum()) > 0.75 * len(evs)
    assert np.array_equal(depth, ref["depth"]) and np.array_equal(bgr, ref["bgr"])
    m = ref["mask"]
    est = depth[gt["proj_v"][m], gt["proj_u"][m]]
    ok = est > 0
    rel = np.abs(est[ok] - gt["z_rect"][m][ok]) / gt["z_rect"][m][ok]
    assert ok.mean() > 0.99 and np.median(rel) < 0.01 and np.percentile(rel, 95) < 0.02
    # a denser frame of the same scene goes through the tiled kernel (LDS windows on rotated, distorted geometry)
    evs2, gt2 = rig.render_events(cp, tb, row_stride=3, seed=1)
    ref2 = _ref(tb, evs2)
    with XMapsEngine(tb, assume_time_sorted=True) as eng:
        depth2, bgr2, st2 = eng.process_events(evs2)
    assert st2.n_unsorted == 0 and np.array_equal(depth2, ref2["depth"]) and np.array_equal(bgr2, ref2["bgr"])


def test_pipe_accepts_the_reference_calibration_yaml(tmp_path, golden_dir):
    """RuntimeParams.calib = a calibration YAML in the reference's format (python/cam_proj_calibration.py:10-28,77-108):
    the pipe builds its tables itself (cv2-free builder + GPU X-map) and produces the same frame as the oracle run on
    those tables."""
    import os
    import yaml
    from x_maps_amd import rig
    g = np.load(os.path.join(golden_dir, "g6_esl_calib.npz"))

    def node(a):
        a = np.asarray(a, dtype=float)
        a = a.reshape(a.shape[0], -1)
        return {"type-id": "opencv_matrix", "rows": int(a.shape[0]), "cols": int(a.shape[1]), "dt": "d", "data": a.ravel().tolist()}

    doc = {"camera_intrinsic_matrix": node(g["camera_K"]), "camera_distortion_coefficients": node(rig.NEBRA_CAMERA_D.reshape(1, 5)),
           "projector_intrinsic_matrix": node(g["projector_K"]), "projector_distortion_coefficients": node(g["projector_D"]),
           "relative_rotation": node(g["R"]), "relative_translation": node(g["T"]), "F": node(np.eye(3))}
    path = tmp_path / "calib.yaml"
    path.write_text(yaml.safe_dump(doc))
    params = RuntimeParams(camera_width=640, camera_height=480, projector_width=1080, projector_height=1920, projector_fps=60,
                           z_near=0.1, z_far=1.2, calib=str(path), projector_time_map=None, no_frame_dropping=True,
                           camera_perspective=False)
    got = []
    pipe = DepthReprojectionPipe(params, StatsPrinter(), got.append)
    cp = rig.esl_like_params(os.path.join(golden_dir, "g6_esl_calib.npz"))
    evs, _ = rig.render_events(cp, pipe.calib_maps.tables | {"R1": np.eye(3)}, row_stride=29)
    pipe.process_ev_frame(evs)
    assert got[0].shape == (1920, 1080, 3)
    tb = dict(pipe.calib_maps.tables)
    assert np.array_equal(got[0], _ref(tb, evs)["bgr"])
    pipe.close()


def test_pinned_host_asynchronous_path():
    """XM_MEM_HOST_PINNED: pinned host SoA / AoS in, pinned host depth + BGR out, asynchronous over 3 slots."""
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    with XMapsEngine(tb, n_slots=3) as eng:
        jobs = []
        for f in range(7):
            evs = S.make_events(cfg, frame=40 + f, n=3000 + 111 * f)
            x, y, t, _ = S.to_soa(evs)
            px, py, pt = eng.host_empty(x.shape, np.uint16), eng.host_empty(y.shape, np.uint16), eng.host_empty(t.shape, np.int64)
            px[:], py[:], pt[:] = x, y, t
            pa = eng.host_empty(evs.shape, S.EVENT_CD_DTYPE)
            pa[:] = evs
            d1, b1 = eng.host_empty((cfg.proj_h, cfg.proj_w), np.float32), eng.host_empty((cfg.proj_h, cfg.proj_w, 3), np.uint8)
            d2 = eng.host_empty((cfg.proj_h, cfg.proj_w), np.float32)
            eng.process_frame_pinned(px, py, pt, None, d1, b1)
            eng.process_events_pinned(pa, d2, None)
            jobs.append((evs, d1, b1, d2))
        eng.sync()
        for evs, d1, b1, d2 in jobs:
            ref = _ref(tb, evs)
            assert np.array_equal(d1, ref["depth"]) and np.array_equal(b1, ref["bgr"]) and np.array_equal(d2, ref["depth"])


def test_device_side_pause_detection_matches_numpy(golden_dir):
    """N2: np.nonzero(np.diff(t) >= 40)[0] on the GPU (host SoA, host EventCD, device pointer), incl. the golden G5 stream."""
    import os
    torch = pytest.importorskip("torch")
    g = np.load(os.path.join(golden_dir, "g5_trigger.npz"))
    rng = np.random.default_rng(3)
    streams = [g["t"].astype(np.int64), np.cumsum(rng.integers(0, 60, 200_003)).astype(np.int64),
               np.array([5], np.int64), np.array([5, 100], np.int64), np.zeros(0, np.int64)]
    with XMapsEngine(S.make_tables(S.C_TINY)) as eng:
        for t in streams:
            ref = np.nonzero(np.diff(t) >= 40)[0]
            assert np.array_equal(eng.find_pauses(t=t), ref)
            ev = np.zeros(len(t), S.EVENT_CD_DTYPE)
            ev["t"] = t
            assert np.array_equal(eng.find_pauses(evs=ev), ref)
            if len(t):
                dt = torch.from_numpy(t).to("cuda:0")
                torch.cuda.synchronize()
                assert np.array_equal(eng.find_pauses(device_ptr=dt.data_ptr(), n=len(t)), ref)
        assert np.array_equal(eng.find_pauses(t=streams[1], thresh_us=55), np.nonzero(np.diff(streams[1]) >= 55)[0])


def test_try_sorted_mode_redoes_unsorted_frames_asynchronously():
    """XM_FLAG_TRY_SORTED: no declaration from the caller.  Asynchronous device-pointer frames: sorted ones take the
    shortcut, shuffled ones are redone on the general path when their slot comes round again or in xm_sync; every
    output is exact."""
    import torch
    dev = torch.device("cuda", 0)
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(11)
    frames, refs = [], []
    for f in range(7):
        ev = S.make_events(cfg, frame=f, n=30_000)
        x, y, t, _ = S.to_soa(ev)
        if f in (1, 2, 5):  # not sorted: first / last event are not the extrema
            perm = rng.permutation(len(t))
            x, y, t = x[perm], y[perm], t[perm]
        refs.append(O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t))
        frames.append(tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)))
    for n_slots in (1, 2, 3):
        with XMapsEngine(tb, n_slots=n_slots, try_sorted=True) as eng:
            outs = [torch.zeros((eng.out_h, eng.out_w), dtype=torch.float32, device=dev) for _ in frames]
            bgrs = [torch.zeros((eng.out_h, eng.out_w, 3), dtype=torch.uint8, device=dev) for _ in frames]
            torch.cuda.synchronize()
            for (fx, fy, ft), o, b in zip(frames, outs, bgrs):
                eng.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, len(ft), o.data_ptr(), b.data_ptr())
            eng.sync()
            assert eng.sorted_fallbacks() == 3
            for o, b, r in zip(outs, bgrs, refs):
                assert np.array_equal(o.cpu().numpy(), r["depth"]) and np.array_equal(b.cpu().numpy(), r["bgr"])
            # synchronous host call in the same mode: redone inside the call
            x, y, t = (a.cpu().numpy() for a in frames[1])
            d, b, st = eng.process_frame(x.view(np.uint16), y.view(np.uint16), t)
            assert st.n_unsorted > 0 and np.array_equal(d, refs[1]["depth"]) and eng.sorted_fallbacks() == 4
            eng.sync()


def test_try_sorted_mode_pinned_host_path_and_aos():
    """XM_FLAG_TRY_SORTED through the asynchronous pinned-host path (inputs staged per slot, outputs copied back) and with
    AoS EventCD records: shuffled frames are redone (staging buffers still hold them) and the host outputs are exact."""
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(3)
    with XMapsEngine(tb, n_slots=2, try_sorted=True) as eng:
        jobs = []
        for f in range(6):
            ev = S.make_events(cfg, frame=f, n=25_000)
            if f % 2:
                ev = ev[rng.permutation(len(ev))]
            x, y, t, _ = S.to_soa(ev)
            ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
            d = eng.host_empty((eng.out_h, eng.out_w), np.float32)
            b = eng.host_empty((eng.out_h, eng.out_w, 3), np.uint8)
            if f < 4:
                px, py, pt = eng.host_empty(x.shape, np.uint16), eng.host_empty(y.shape, np.uint16), eng.host_empty(t.shape, np.int64)
                px[:], py[:], pt[:] = x, y, t
                eng.process_frame_pinned(px, py, pt, None, d, b)
                jobs.append((d, b, ref, (px, py, pt)))
            else:
                pe = eng.host_empty(ev.shape, ev.dtype)
                pe[:] = ev
                eng.process_events_pinned(pe, d, b)
                jobs.append((d, b, ref, pe))
        eng.sync()
        assert eng.sorted_fallbacks() == 3
        for d, b, ref, _ in jobs:
            assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])


def test_launch_workers_keep_order_and_results():
    """XM_FLAG_LAUNCH_WORKERS: asynchronous device frames are launched by one worker thread per slot stream.  Interleaved
    with synchronous host calls, profile calls, stage calls and try-sorted redos the results stay exact."""
    import torch
    dev = torch.device("cuda", 0)
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    rng = np.random.default_rng(23)
    frames, refs = [], []
    for f in range(9):
        ev = S.make_events(cfg, frame=f, n=28_000)
        x, y, t, _ = S.to_soa(ev)
        if f % 4 == 1:
            perm = rng.permutation(len(t))
            x, y, t = x[perm], y[perm], t[perm]
        refs.append(O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t))
        frames.append((x, y, t) + tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)))
    for try_sorted in (False, True):
        with XMapsEngine(tb, n_slots=3, try_sorted=try_sorted, launch_workers=True) as eng:
            outs = [torch.zeros((eng.out_h, eng.out_w), dtype=torch.float32, device=dev) for _ in frames]
            bgrs = [torch.zeros((eng.out_h, eng.out_w, 3), dtype=torch.uint8, device=dev) for _ in frames]
            torch.cuda.synchronize()
            for rep in range(3):
                for i, (x, y, t, fx, fy, ft) in enumerate(frames):
                    eng.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, len(t), outs[i].data_ptr(), bgrs[i].data_ptr())
                    if i == 4:  # a synchronous host call and a stage call in the middle of the asynchronous stream
                        d, b, st = eng.process_frame(x, y, t)
                        assert np.array_equal(d, refs[i]["depth"]) and st.n_inliers == int(refs[i]["mask"].sum())
                        xr, yr = eng.rectify_cam_coords_i16(x[:100], y[:100])
                        assert np.array_equal(xr, tb["cam_mapx_i16"][y[:100], x[:100]])
                    if i == 6:
                        st = eng.profile_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, len(t), outs[i].data_ptr(), bgrs[i].data_ptr())
                        assert st.n_inliers == int(refs[i]["mask"].sum())
                eng.sync()
                for o, b, r in zip(outs, bgrs, refs):
                    assert np.array_equal(o.cpu().numpy(), r["depth"]) and np.array_equal(b.cpu().numpy(), r["bgr"])
            if try_sorted:
                assert eng.sorted_fallbacks() >= 6
