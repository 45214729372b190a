"""-m gpu: device-side ingest (xm_ingest_*): raw packets -> polarity / activity filter -> pause detection -> frame cut ->
K0/K1/K2 on the cut frame, all on the device, against the CPU chain oracle/ingest_oracle.py (+ the hot-path oracle) and the
reference's own trigger-finder run (golden G5)."""
import os

import numpy as np
import pytest

from conftest import xm_option

import ingest_oracle as IO
import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S
from x_maps_amd.ingest import DeviceIngest

pytestmark = pytest.mark.gpu


def _packets(stream, packet_us):
    edges = np.arange(stream["t"][0], stream["t"][-1] + packet_us, packet_us)
    cuts = np.searchsorted(stream["t"], edges)
    return [stream[a:b] for a, b in zip(cuts[:-1], cuts[1:])]


def _tiny_stream(n_frames, seed, per_frame=2600, neg=0.1, gap_noise=3):
    cfg = S.C_TINY
    rng = np.random.default_rng(seed)
    chunks = []
    for f in range(n_frames):
        start = 2_000_000 + f * 16_600
        tt = np.unique(np.concatenate((np.sort(rng.integers(0, 13_000, per_frame)) + start, np.arange(start, start + 13_000, 25))))
        ev = np.zeros(len(tt), S.EVENT_CD_DTYPE)
        ev["t"] = tt
        ev["x"] = np.clip((tt - start) / 13_000 * cfg.cam_w + rng.normal(0, 1.5, len(tt)), 0, cfg.cam_w - 1).astype(np.uint16)
        ev["y"] = rng.integers(0, cfg.cam_h, len(tt))
        ev["p"] = rng.random(len(tt)) >= neg
        parts = [ev]
        if gap_noise and f % gap_noise == gap_noise - 1:
            nz = np.zeros(1, S.EVENT_CD_DTYPE)
            nz["t"], nz["x"], nz["y"], nz["p"] = start + 14_500, 5, 5, 1
            parts.append(nz)
        chunks.append(np.concatenate(parts))
    return np.concatenate(chunks)


def _check_frames(tb, got, want_frames, camera=False):
    assert len(got) == len(want_frames), (len(got), len(want_frames))
    for fr, evs in zip(got, want_frames):
        assert (fr.n_events, fr.t_first, fr.t_last) == (len(evs), int(evs["t"][0]), int(evs["t"][-1])), (fr.seq, fr.lost, fr.overflow)
        x, y, t, _ = S.to_soa(evs)
        ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera)
        assert fr.n_inliers == int(ref["mask"].sum()) and fr.n_index_errors == 0 and not fr.lost and fr.overflow == 0
        assert np.array_equal(fr.depth, ref["depth"]) and np.array_equal(fr.bgr, ref["bgr"]), fr.seq


def test_golden_trigger_stream_is_cut_on_the_device_like_the_reference(golden_dir):
    """The reference's own RobustTriggerFinder run (golden G5: packets, frames' first/last t and lengths) reproduced by the
    device-side segmentation; every frame's depth/BGR == oracle on those events."""
    g = np.load(os.path.join(golden_dir, "g5_trigger.npz"))
    ev = np.zeros(len(g["t"]), S.EVENT_CD_DTYPE)
    ev["x"], ev["y"], ev["t"], ev["p"] = g["x"], g["y"], g["t"], 1
    tb = S.make_tables(S.C_TINY)
    with XMapsEngine(tb) as eng, DeviceIngest(eng, int(g["fps"]), capacity_events=1 << 16, max_packet_events=1 << 13, result_ring=64) as ing:
        got = []
        cuts = g["packet_cuts"]
        for a, b in zip(cuts[:-1], cuts[1:]):
            ing.push(ev[a:b])
            got += ing.poll()
        ing.flush()
        got += ing.poll()
    assert len(got) == int(g["n_frames"])
    assert [f.n_events for f in got] == list(g["frame_len"])
    assert [f.t_first for f in got] == list(g["frame_first_t"]) and [f.t_last for f in got] == list(g["frame_last_t"])
    tf = IO.TriggerFinderOracle(int(g["fps"]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        tf.process_events(ev[a:b])
    _check_frames(tb, got, tf.frames)


@pytest.mark.parametrize("camera", [False, True])
@pytest.mark.parametrize("activity", [False, True])
def test_filters_and_segmentation_match_the_cpu_chain(camera, activity):
    """Polarity filter + (own-definition) activity filter + segmentation on the device == the sequential CPU chain, on a
    stream with negative events, gap noise and a buffer small enough to force several compactions."""
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(14, seed=7 + activity)
    pk = _packets(stream, int(1e6 / 60 / 4))
    tf = IO.TriggerFinderOracle(60)
    act = IO.ActivityFilterOracle(S.C_TINY.cam_w, S.C_TINY.cam_h, int(1e6 / 60))
    for p in pk:
        pos = IO.polarity_filter(p)
        tf.process_events(act.process(pos) if activity else pos)
    assert len(tf.frames) >= 4  # the reference finder loses lock easily (a buffer with a single pause is dropped): by design
    with XMapsEngine(tb, camera_perspective=camera) as eng, \
            DeviceIngest(eng, 60, activity_filter=activity, capacity_events=1 << 13, max_packet_events=1 << 11, result_ring=64) as ing:
        got = []
        for p in pk:
            ing.push(p)
            got += ing.poll()
        ing.flush()
        got += ing.poll()
    _check_frames(tb, got, tf.frames, camera)


def test_activity_filter_with_packets_longer_than_its_threshold():
    """Packets spanning several thresholds are split into sub-packets on the way in; the result is the same rule."""
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(10, seed=21)
    tf = IO.TriggerFinderOracle(60)
    act = IO.ActivityFilterOracle(S.C_TINY.cam_w, S.C_TINY.cam_h, 2_000)
    pk = _packets(stream, int(1e6 / 60 / 4))
    for p in pk:
        tf.process_events(act.process(IO.polarity_filter(p)))
    assert len(tf.frames) >= 2
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, activity_filter=True, activity_thresh_us=2_000,
                                              capacity_events=1 << 14, max_packet_events=1 << 12, result_ring=64) as ing:
        got = []
        for p in pk:  # 4.2 ms packets against a 2 ms threshold
            ing.push(p)
        ing.flush()
        got += ing.poll()
    _check_frames(tb, got, tf.frames)


def test_esl_like_stream_stays_on_the_device():
    """BASELINE config 3 stand-in through the device-side ingest: ~150 k events / frame, negative events, gap noise; frames
    go through the one-thread-per-event K1 first and the tiled one once the stream's density is known."""
    from x_maps_amd import rig
    cp, tb, _, _ = rig.make_esl_like(row_stride=13)
    stream, _ = rig.render_stream(cp, tb, n_frames=12, row_stride=13, seed=5)
    pk = _packets(stream, int(1e6 / 60 / 4))
    tf = IO.TriggerFinderOracle(60)
    for p in pk:
        tf.process_events(IO.polarity_filter(p))
    assert len(tf.frames) >= 4
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, capacity_events=1 << 20, max_packet_events=1 << 17, result_ring=64) as ing:
        got = []
        for p in pk:
            ing.push(p)
            got += ing.poll()
        ing.flush()
        got += ing.poll()
    _check_frames(tb, got, tf.frames)


def test_pipe_with_device_ingest_calls_back_with_the_same_frames():
    """DepthReprojectionProcessor(params) with device_ingest=True: process_events(packet) pushes the RAW packet; the frames
    handed to the window equal those of the host path (polarity filter + host trigger finder + fused frame)."""
    from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor, RuntimeParams
    cfg = S.C_TINY
    tb = S.make_tables(cfg)
    stream = _tiny_stream(9, seed=3)
    pk = _packets(stream, int(1e6 / 60 / 4))

    def run(device_ingest):
        shown = []

        class Window:
            def should_close(self):
                return False

            def show_async(self, img):
                shown.append(img)

        params = RuntimeParams(camera_width=cfg.cam_w, camera_height=cfg.cam_h, projector_width=cfg.proj_w,
                               projector_height=cfg.proj_h, projector_fps=60, z_near=0.1, z_far=1.2, calib=None,
                               projector_time_map=None, no_frame_dropping=True, camera_perspective=False, tables=tb,
                               device_ingest=device_ingest)
        with DepthReprojectionProcessor(params, window=Window()) as proc:
            for p in pk:
                proc.process_events(p)
            proc.flush()
        return shown

    a, b = run(False), run(True)
    assert len(a) == len(b) >= 3
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def _processor_params(tb, cfg=S.C_TINY, **kw):
    from x_maps_amd.depth_reprojection_processor import RuntimeParams
    return RuntimeParams(camera_width=cfg.cam_w, camera_height=cfg.cam_h, projector_width=cfg.proj_w, projector_height=cfg.proj_h,
                         projector_fps=60, z_near=0.1, z_far=1.2, calib=None, projector_time_map=None, no_frame_dropping=True,
                         camera_perspective=False, tables=tb, **kw)


def test_default_params_take_the_device_ingest_and_hand_out_frames_of_the_consumers_own():
    """The reference's call pattern -- `with DepthReprojectionProcessor(params)` + process_events(packet),
    depth_reprojection_processor.py:66-69,107-111 -- with DEFAULT RuntimeParams: the packets go to the device ingest; the window
    gets the same frames as from the host chain (polarity mask, activity filter, NumPy trigger finder, one fused call per
    frame), and every frame stays intact however long the window keeps it (a result ring of 4 is lapped several times here):
    the reference's fresh-array contract, SURVEY 8(b) "Ownership"."""
    from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(14, seed=4)
    pk = _packets(stream, int(1e6 / 60 / 4))

    def run(**kw):
        shown = []

        class Window:
            def should_close(self):
                return False

            def show_async(self, img):
                shown.append(img)  # (kept: never copied)

        with DepthReprojectionProcessor(_processor_params(tb, **kw), window=Window()) as proc:
            assert (proc._pipe.ingest is not None) == kw.get("device_ingest", True)
            for p in pk:
                proc.process_events(p)
            proc.flush()
        return shown

    want = run(device_ingest=False)
    got = run(ingest_result_ring=4)  # defaults otherwise
    assert len(got) == len(want) >= 8
    for x, y in zip(got, want):
        assert x.flags.writeable and np.array_equal(x, y)  # (read after the processor, its ingest and its engine are gone)
    snap = [g.copy() for g in got]
    del want
    for g, c in zip(got, snap):
        assert np.array_equal(g, c)


def test_owned_result_buffers_return_to_the_pool_and_are_reused():
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(12, seed=8)
    pk = _packets(stream, int(1e6 / 60 / 4))
    tf = IO.TriggerFinderOracle(60)
    for p in pk:
        tf.process_events(IO.polarity_filter(p))
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, capacity_events=1 << 13, max_packet_events=1 << 11, result_ring=4, lossless=True) as ing:
        held = []
        for p in pk:
            ing.push(p)
            held += ing.poll()  # copy=True: every frame's buffers are the caller's
        ing.flush()
        held += ing.poll()
        _check_frames(tb, held, tf.frames)  # all of them intact although the ring of 4 went round three times
        st = ing.pool_stats()
        assert st["outstanding"] == 2 * len(held) == st["allocated"], st  # (depth + BGR per frame)
        base = held[0].bgr.base
        view = held[0].bgr[::2, ::2]  # a view keeps the buffer alive ...
        del held, base
        import gc
        gc.collect()
        st2 = ing.pool_stats()
        assert st2["outstanding"] == 1 and st2["spare"] == st["outstanding"] - 1, st2
        del view
        gc.collect()
        assert ing.pool_stats()["outstanding"] == 0
        # ... and a second pass over the stream makes no new buffer: the pool's spare ones go round
        ing.reset()
        n_alloc = ing.pool_stats()["allocated"]
        again = []
        for p in pk:
            ing.push(p)
            again += ing.poll()
        ing.flush()
        again += ing.poll()
        _check_frames(tb, again, tf.frames)
        assert ing.pool_stats()["allocated"] == n_alloc
        keep = again[3].bgr  # outlives the ingest: released into a closed pool, freed there
        want = keep.copy()
    assert np.array_equal(keep, want)
    del keep, again


def test_an_exhausted_pool_falls_back_to_copies():
    xm_option("XM_INGEST_POOL_CAP", "3")
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(8, seed=9)
    pk = _packets(stream, int(1e6 / 60 / 4))
    tf = IO.TriggerFinderOracle(60)
    for p in pk:
        tf.process_events(IO.polarity_filter(p))
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, capacity_events=1 << 13, max_packet_events=1 << 11, result_ring=16) as ing:
        for p in pk:
            ing.push(p)
        ing.flush()
        got = ing.poll()
        assert ing.pool_stats()["allocated"] <= 3
        _check_frames(tb, got, tf.frames)
        assert got[0].bgr.base is not None and got[2].bgr.base is None and got[2].bgr.flags.owndata  # (the pool's buffer / a plain copy)


def test_a_frame_event_filter_moves_the_stream_to_the_host_chain_and_back():
    """Key E (select_next_frame_event_filter, pipe:169): the frame event filters work on the cut frame's events on the host
    (pipe:131-139), so while one is selected the packets take the host chain; back on NoFilter they take the ingest again."""
    from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor
    from x_maps_amd.frame_event_filter import NoFilter
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(30, seed=6)
    pk = _packets(stream, int(1e6 / 60 / 4))
    third = len(pk) // 3
    shown = []

    class Window:
        def should_close(self):
            return False

        def show_async(self, img):
            shown.append(img)

    with DepthReprojectionProcessor(_processor_params(tb), window=Window()) as proc:
        pipe = proc._pipe
        for p in pk[:third]:
            proc.process_events(p)
        proc.flush()
        n0 = len(shown)
        assert n0 >= 2 and not pipe._host_chain_active
        proc.keyboard_cb("e", None, "release")
        assert not isinstance(pipe.ev_filter_proc.selected_filter(), NoFilter)
        for p in pk[third:2 * third]:
            proc.process_events(p)
        n1 = len(shown)
        assert n1 > n0 and pipe._host_chain_active and pipe.stats_printer.metrics["frame evs filtered out [%]"].max > 0.0
        while not isinstance(pipe.ev_filter_proc.selected_filter(), NoFilter):
            proc.keyboard_cb("e", None, "release")
        for p in pk[2 * third:]:
            proc.process_events(p)
        proc.flush()
        assert len(shown) > n1 and not pipe._host_chain_active


def test_the_slot_is_cleared_before_its_frame_tag_can_wrap(monkeypatch):
    """The ingest slot's tag advances on the device (one per cut frame) and is 19 bits wide in the packed keys: the host clears
    the slot every KEY_MAX_TAG - 16 pushes at the latest.  XM_INGEST_CLEAR_EVERY=3 makes that happen every third push here: the
    frames must come out exactly as without it (every clear lands between two frames of the stream)."""
    xm_option("XM_INGEST_CLEAR_EVERY", "3")
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(14, seed=31)
    pk = _packets(stream, int(1e6 / 60 / 4))
    tf = IO.TriggerFinderOracle(60)
    for p in pk:
        tf.process_events(IO.polarity_filter(p))
    assert len(tf.frames) >= 4
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, capacity_events=1 << 13, max_packet_events=1 << 11, result_ring=64) as ing:
        got = []
        for p in pk:
            ing.push(p)
            got += ing.poll()
        ing.flush()
        got += ing.poll()
    _check_frames(tb, got, tf.frames)


def test_min_events_per_frame_below_four_is_rejected():
    """The frame is evs[prev + 2 : next - 2] (trigger_finder.py:172): fewer than 4 events between two pauses cannot be a frame."""
    from x_maps_amd._native import XMapsNativeError
    tb = S.make_tables(S.C_TINY)
    with XMapsEngine(tb) as eng:
        with pytest.raises((XMapsNativeError, ValueError)):
            DeviceIngest(eng, 60, min_events_per_frame=2, result_ring=64)


@pytest.mark.parametrize("launch_thread", [True, False])
def test_the_ring_wraps_many_times(launch_thread):
    """40 frames through a ring of 8192 events (~13 wrap-arounds; frames that straddle the ring's end are read through the
    mirrored head), launches from the ingest's own thread and from the caller: the same frames as the CPU chain"""
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(40, seed=11)
    pk = _packets(stream, int(1e6 / 60 / 4))
    tf = IO.TriggerFinderOracle(60)
    for p in pk:
        tf.process_events(IO.polarity_filter(p))
    assert len(tf.frames) >= 12
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, capacity_events=1 << 13, max_packet_events=1 << 11, result_ring=4,
                                              launch_thread=launch_thread) as ing:
        got = []
        for p in pk:
            ing.push(p)
            ing.flush()  # (a result ring of 4: pick every frame up before it can be lapped)
            got += ing.poll()
        hs = ing.host_stats()
    assert hs["pushes"] == len(pk)
    _check_frames(tb, got, tf.frames)


def test_packets_of_any_size_and_empty_packets():
    """the same stream in packets of 1 .. 7000 events (several blocks of 2048 per packet, blocks that keep nothing, pauses at
    block and packet borders) with empty pushes in between: the trigger finder sees the same buffer at every decision only if
    the packets are the same, so the CPU chain gets the very same packets"""
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(16, seed=19)
    rng = np.random.default_rng(4)
    cuts = [0]
    while cuts[-1] < len(stream):
        cuts.append(min(len(stream), cuts[-1] + int(rng.choice([1, 2, 63, 64, 65, 500, 2047, 2048, 2049, 4100, 7000]))))
    pk = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        pk.append(stream[a:b])
        if rng.random() < 0.2:
            pk.append(stream[:0])
    # a run of negative events as long as two blocks: blocks that keep nothing
    run = stream[cuts[5]:cuts[5] + 1].repeat(4500)
    run["p"] = 0
    pk.insert(6, run)
    tf = IO.TriggerFinderOracle(60)
    for p in pk:
        tf.process_events(IO.polarity_filter(p))
    assert len(tf.frames) >= 4
    # (the ring must hold what the trigger finder may keep -- up to two periods -- plus a full packet: 1 << 16)
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, capacity_events=1 << 16, max_packet_events=1 << 13, result_ring=64) as ing:
        got = []
        for p in pk:
            ing.push(p)
            got += ing.poll()
        ing.flush()
        got += ing.poll()
    _check_frames(tb, got, tf.frames)


def test_a_ring_without_room_drops_its_live_part_and_says_so():
    """a stream without any pause never yields a frame (the reference's buffer would grow without bound); once the live part
    leaves no room for another full packet it is dropped and counted, and the stream behind it is cut as usual"""
    tb = S.make_tables(S.C_TINY)
    n = 20_000
    ev = np.zeros(n, S.EVENT_CD_DTYPE)
    ev["t"] = 1_000_000 + np.arange(n) // 4  # 4 events per us for 5 ms: no pause, less than a period
    ev["x"], ev["y"], ev["p"] = 3, 3, 1
    tail = _tiny_stream(8, seed=2)
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, capacity_events=1 << 13, max_packet_events=1 << 12, result_ring=64) as ing:
        for a in range(0, n, 4000):
            ing.push(ev[a:a + 4000])  # 4000 live: room; 8000 live: no room for 4096 more -> dropped; and again
        for p in _packets(tail, int(1e6 / 60 / 4)):
            ing.push(p)
        ing.flush()
        got = ing.poll()
        ds = ing.device_stats()
    assert len(got) >= 2 and all(f.overflow == 16_000 and not f.lost for f in got)
    assert ds["events_dropped"] == 16_000 and ds["frames_cut"] == len(got) and ds["events_appended"] > 0
    x, y, t, _ = S.to_soa(IO.polarity_filter(tail))
    for f in got:  # every frame is a contiguous piece of the tail stream's positive events, processed like any other
        a = int(np.searchsorted(t, f.t_first))
        while not (t[a + f.n_events - 1] == f.t_last):
            a += 1
        ref = O.process_ev_frame(tb, x[a:a + f.n_events].astype(np.int64), y[a:a + f.n_events].astype(np.int64), t[a:a + f.n_events])
        assert np.array_equal(f.depth, ref["depth"])


def test_views_into_the_result_ring():
    """poll(copy=False) hands out views into the pinned ring: same pixels as the copies, same buffers coming round again"""
    tb = S.make_tables(S.C_TINY)
    stream = _tiny_stream(14, seed=7)
    pk = _packets(stream, int(1e6 / 60 / 4))
    with XMapsEngine(tb) as e1, XMapsEngine(tb) as e2, DeviceIngest(e1, 60, capacity_events=1 << 14, max_packet_events=1 << 12,
                                                                       result_ring=2) as a, \
            DeviceIngest(e2, 60, capacity_events=1 << 14, max_packet_events=1 << 12, result_ring=64) as b:
        n = 0
        addrs = set()
        for p in pk:
            a.push(p), b.push(p)
            a.flush(), b.flush()
            va, vb = a.poll(copy=False), b.poll()
            assert len(va) == len(vb)
            for x, y in zip(va, vb):
                assert np.array_equal(x.depth, y.depth) and np.array_equal(x.bgr, y.bgr)
                assert x.bgr.base is not None and type(y.bgr.base).__name__ == "_OwnedBuffer"  # (a view into the ring / the caller's own buffer)
                addrs.add(x.bgr.ctypes.data)
                n += 1
        assert n >= 4 and len(addrs) == 2
