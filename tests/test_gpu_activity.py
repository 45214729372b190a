"""-m gpu: the activity-noise filter on the device (k_act_first + the flags inside k_ing_count, or alone through xm_activity_*)
against the sequential definition oracle/ingest_oracle.py:ActivityFilterOracle -- flag by flag, for packets of every shape: one
time bucket, several, more than the parallel path takes, stamps running backwards (the sequential path on the device)."""
import numpy as np
import pytest

import ingest_oracle as IO
import xmaps_oracle as O
from x_maps_amd import XMapsEngine, evt2, evt3
from x_maps_amd import synthetic as S
from x_maps_amd.activity_filter import ActivityNoiseFilterAlgorithm
from x_maps_amd.ingest import DeviceIngest

import test_gpu_ingest as TI

pytestmark = pytest.mark.gpu
CFG = S.C_TINY


def _stream(n, seed, span_us, cluster=0.6, start=1_000_000, sort=True):
    """n events over span_us: a share of them in clusters (neighbours in space and time), the rest uniform noise"""
    rng = np.random.default_rng(seed)
    ev = np.zeros(n, S.EVENT_CD_DTYPE)
    t = rng.integers(0, max(1, span_us), n)
    if sort:
        t = np.sort(t)
    ev["t"] = start + t
    cx, cy = rng.integers(0, CFG.cam_w, n), rng.integers(0, CFG.cam_h, n)
    near = rng.random(n) < cluster
    # clustered events sit next to the previous event's pixel
    x, y = cx.copy(), cy.copy()
    for i in range(1, n):
        if near[i]:
            x[i] = min(max(x[i - 1] + rng.integers(-1, 2), 0), CFG.cam_w - 1)
            y[i] = min(max(y[i - 1] + rng.integers(-1, 2), 0), CFG.cam_h - 1)
    ev["x"], ev["y"], ev["p"] = x, y, 1
    return ev


def _check(eng, packets, thresh, want_sequential=None, max_packet=0, include_self=False):
    ora = IO.ActivityFilterOracle(CFG.cam_w, CFG.cam_h, thresh, include_self=include_self)
    with ActivityNoiseFilterAlgorithm(eng, thresh, max_packet_events=max_packet, include_self=include_self) as act:
        for k, p in enumerate(packets):
            want = ora.process(p)
            got = act.process_events(p)
            assert len(got) == len(want) and np.array_equal(got, want), (k, len(p), len(got), len(want))
        seq = act.sequential_packets()
    if want_sequential is not None:
        assert (seq > 0) == want_sequential, seq
    return seq


@pytest.fixture(scope="module")
def eng():
    with XMapsEngine(S.make_tables(CFG)) as e:
        yield e


def test_single_bucket_packets(eng):
    ev = _stream(6000, 1, 40_000)
    pk = TI._packets(ev, 4_000)
    _check(eng, pk, 16_666, want_sequential=False)


@pytest.mark.parametrize("shape", ["one bucket", "several buckets", "stamps running backwards"])
def test_the_variant_whose_window_includes_the_own_pixel(eng, shape):
    """XM_INGEST_ACT_SELF / xm_activity_set_rule: an earlier event at the event's OWN pixel qualifies too (one of the ways Metavision's
    filter may differ from this build's definition: oracle/ingest_oracle.py) -- parallel and sequential paths == the oracle's variant,
    and the variant does keep events the default rule drops (a pixel firing repeatedly)"""
    if shape == "one bucket":
        ev, span, T, seq = _stream(6000, 11, 40_000, cluster=0.3), 4_000, 16_666, False
    elif shape == "several buckets":
        ev, span, T, seq = _stream(9000, 12, 60_000, cluster=0.3), 9_000, 2_000, False
    else:
        ev, span, T, seq = _stream(5000, 13, 40_000, cluster=0.3, sort=False), 10 ** 9, 3_000, True
    rng = np.random.default_rng(14)
    hot = rng.random(len(ev)) < 0.2  # a fifth of the events at three hot pixels far from everything else's neighbourhood
    ev["x"][hot], ev["y"][hot] = rng.integers(0, 3, int(hot.sum())) * 7 + 2, 1
    pk = TI._packets(ev, span) if span < 10 ** 9 else [ev[i:i + 1200] for i in range(0, len(ev), 1200)]
    _check(eng, pk, T, want_sequential=seq, include_self=True)
    a, b = IO.ActivityFilterC(CFG.cam_w, CFG.cam_h, T), IO.ActivityFilterC(CFG.cam_w, CFG.cam_h, T, include_self=True)
    assert sum(len(b.process(p)) for p in pk) > sum(len(a.process(p)) for p in pk)


def test_packets_spanning_several_thresholds_stay_parallel(eng):
    """2 ms threshold, 9 ms packets: 5 buckets per packet; the cells of bucket b - 1 decide by stamp, those of b by index"""
    ev = _stream(9000, 2, 60_000)
    pk = TI._packets(ev, 9_000)
    _check(eng, pk, 2_000, want_sequential=False)


def test_threshold_boundaries_are_exact(eng):
    """pairs of neighbouring events exactly T, T + 1 and T - 1 apart, across and inside buckets and packets"""
    T = 1000
    rows = []
    t = 5_000
    for k, d in enumerate([T - 1, T, T + 1, 0, 1, 2 * T, T, T + 1, T]):
        x = 3 + 4 * k
        rows += [(x, 5, t), (x + 1, 6, t + d)]
        t += 137
    ev = np.zeros(len(rows), S.EVENT_CD_DTYPE)
    order = np.argsort([r[2] for r in rows], kind="stable")
    for i, j in enumerate(order):
        ev["x"][i], ev["y"][i], ev["t"][i] = rows[j]
    ev["p"] = 1
    _check(eng, [ev], T, want_sequential=False)
    _check(eng, [ev[:7], ev[7:]], T, want_sequential=False)
    _check(eng, [ev[i:i + 1] for i in range(len(ev))], T, want_sequential=False)


def test_more_buckets_than_the_parallel_path_takes(eng):
    ev = _stream(5000, 3, 30_000)
    seq = _check(eng, [ev[:2500], ev[2500:]], 1_000, want_sequential=True)  # 15 ms per packet against 8 x 1.001 ms
    assert seq == 2


@pytest.mark.parametrize("seed", [4, 5, 6])
def test_stamps_running_backwards_take_the_sequential_path(eng, seed):
    ev = _stream(4000, seed, 50_000, sort=False)
    _check(eng, [ev[:1500], ev[1500:1501], ev[1501:]], 5_000, want_sequential=True)


def test_small_disorder_inside_one_bucket_stays_parallel(eng):
    """bucket numbers, not stamps, must run forwards: jitter well inside a bucket does not leave the parallel path"""
    ev = _stream(3000, 7, 3_000)
    rng = np.random.default_rng(7)
    ev["t"][1:] += rng.integers(-40, 40, len(ev) - 1)
    ev["t"][0] = ev["t"].min() - 1
    _check(eng, [ev], 16_666, want_sequential=False)


def test_history_carries_over_packets_and_reset(eng):
    ev = _stream(3000, 8, 20_000)
    ora = IO.ActivityFilterOracle(CFG.cam_w, CFG.cam_h, 4_000)
    with ActivityNoiseFilterAlgorithm(eng, 4_000) as act:
        for p in TI._packets(ev, 1_000):
            assert np.array_equal(act.process_events(p), ora.process(p))
        act.reset()
        ora = IO.ActivityFilterOracle(CFG.cam_w, CFG.cam_h, 4_000)
        p = ev[:500]
        assert np.array_equal(act.process_events(p), ora.process(p))


def test_long_packets_go_through_in_pieces(eng):
    ev = _stream(7000, 9, 12_000)
    _check(eng, [ev], 16_666, max_packet=2048)


def test_mask_form_and_empty_packet(eng):
    ev = _stream(500, 10, 2_000)
    with ActivityNoiseFilterAlgorithm(eng, 16_666) as act:
        assert len(act.process_events(ev[:0])) == 0
        m = act.process_events(ev, return_mask=True)
    ora = IO.ActivityFilterOracle(CFG.cam_w, CFG.cam_h, 16_666)
    want = ora.process(ev)
    assert m.dtype == bool and np.array_equal(ev[m], want)


# ---- through the ingest: every kind of packet, flags consumed on the device -----------------------------------------------------
def _frames_cpu(pk, thresh=int(1e6 / 60), include_self=False):
    tf = IO.TriggerFinderOracle(60)
    act = IO.ActivityFilterOracle(CFG.cam_w, CFG.cam_h, thresh, include_self=include_self)
    for p in pk:
        tf.process_events(act.process(IO.polarity_filter(p)))
    return tf.frames


@pytest.mark.parametrize("fmt", [3, 2])
@pytest.mark.parametrize("count", [False, True])
def test_ingest_filter_on_chunks_decoded_on_the_device(fmt, count):
    """EVT 3.0 / 2.0 words -> decoder -> activity filter -> segmentation, nothing waited for (count = False: the chunk's event
    count stays on the device) == the CPU chain on the same packets as records"""
    tb = S.make_tables(CFG)
    stream = TI._tiny_stream(12, seed=31 + fmt)
    pk = TI._packets(stream, int(1e6 / 60 / 4))
    want = _frames_cpu(pk)
    assert len(want) >= 3
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, activity_filter=True, capacity_events=1 << 14, max_packet_events=1 << 12, result_ring=64) as ing:
        dec = evt2.DeviceEvt2Decoder(eng, max_words=1 << 15) if fmt == 2 else evt3.DeviceEvt3Decoder(eng, max_words=1 << 15)
        got = []
        for p in pk:
            w = evt2.encode_evt2(p, time_high_every_us=16) if fmt == 2 else evt3.encode_evt3(p)
            dec.push(ing, w, count=count)
            got += ing.poll()
        ing.flush()
        got += ing.poll()
        assert ing.activity_sequential_packets() == 0
        dec.close()
    TI._check_frames(tb, got, want)


def test_first_pass_rides_on_the_packet_before_when_packets_queue_up_and_not_for_a_packet_alone():
    """round 6: pushed back to back (a replay) the packets' first passes go out inside their predecessors' k_ing_count launch
    (k_ing_count_act); a packet that arrives alone (flush after every push: a live camera's situation) gets a launch of its own.
    Both == the CPU chain, frame by frame; so does a caller that issues the launches itself (no threads: never fused)."""
    tb = S.make_tables(CFG)
    stream = TI._tiny_stream(10, seed=53)
    pk = TI._packets(stream, int(1e6 / 60 / 4))
    want = _frames_cpu(pk)
    assert len(want) >= 2
    for mode in ("queued", "alone", "no threads"):
        with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, activity_filter=True, capacity_events=1 << 15, max_packet_events=1 << 13,
                                                  result_ring=64, launch_thread=mode != "no threads") as ing:
            got = []
            for p in pk:
                ing.push(p)
                if mode == "alone":
                    ing.flush()
                    got += ing.poll()
            ing.flush()
            got += ing.poll()
            fused = ing.activity_fused_first_passes()
        TI._check_frames(tb, got, want)
        if mode == "queued":
            assert fused > 0, (fused, len(pk))  # (how many depends on how far the pushes run ahead of the launch thread)
        else:
            assert fused == 0, (mode, fused)


def test_empty_packets_between_the_packets_do_not_upset_the_turns_of_the_two_sets_of_cells():
    """the packets take the filter's two sets of cells in turns, and the next packet's first pass runs beside this packet's counting
    launch: an empty push in between (a camera's iterator may deliver one) must not make two packets share a set"""
    tb = S.make_tables(CFG)
    stream = TI._tiny_stream(10, seed=59)
    pk = TI._packets(stream, int(1e6 / 60 / 4))
    for every in (1, 2, 3):
        seq = []
        for i, p in enumerate(pk):
            seq.append(p)
            if i % every == 0:
                seq.append(p[:0])
        want = _frames_cpu(seq)  # (the trigger finder decides once per packet: the CPU chain sees the empty ones too)
        assert len(want) >= 2
        with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, activity_filter=True, capacity_events=1 << 15, max_packet_events=1 << 13,
                                                  result_ring=64) as ing:
            for p in seq:
                ing.push(p)
            ing.flush()
            got = ing.poll()
        TI._check_frames(tb, got, want)


def test_ingest_with_the_rule_variants_as_configuration():
    """the strict comparison (threshold - 1) and the own-pixel variant through the ingest's kernels (the fused first pass included:
    the packets are pushed back to back) == the CPU chain with the oracle's variants"""
    tb = S.make_tables(CFG)
    stream = TI._tiny_stream(10, seed=47)
    pk = TI._packets(stream, int(1e6 / 60 / 4))
    T = int(1e6 / 60)
    for own, thr in ((True, T), (False, T - 1), (True, 900)):
        want = _frames_cpu(pk, thr, include_self=own)
        assert len(want) >= 2
        with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, activity_filter=True, activity_thresh_us=thr, activity_include_self=own,
                                                  capacity_events=1 << 15, max_packet_events=1 << 13, result_ring=64) as ing:
            got = []
            for p in pk:
                ing.push(p)
            ing.flush()
            got += ing.poll()
        TI._check_frames(tb, got, want)


def test_ingest_filter_with_period_chunks_and_a_short_threshold():
    """one chunk per projector period (13 ms of events) against a 1 ms threshold: 13 buckets per packet -> the sequential path
    inside the ingest"""
    tb = S.make_tables(CFG)
    stream = TI._tiny_stream(10, seed=41)
    pk = TI._packets(stream, 16_600)
    want = _frames_cpu(pk, 1_000)
    assert len(want) >= 2
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, activity_filter=True, activity_thresh_us=1_000, capacity_events=1 << 15,
                                              max_packet_events=1 << 13, result_ring=64) as ing:
        got = []
        for p in pk:
            ing.push(p)
            got += ing.poll()
        ing.flush()
        got += ing.poll()
        assert ing.activity_sequential_packets() > 0
    TI._check_frames(tb, got, want)


def test_ingest_filter_on_a_stream_with_a_time_glitch():
    """a packet whose stamps step back (a glitch a real reader can produce at a time-base wrap): judged sequentially, the packets
    around it in parallel; the same kept events either way"""
    tb = S.make_tables(CFG)
    stream = TI._tiny_stream(8, seed=43)
    pk = TI._packets(stream, int(1e6 / 60 / 4))
    k = len(pk) // 2
    bad = pk[k].copy()
    bad["t"][len(bad) // 2:] -= 20_000  # (the trigger finder sees a negative diff: no pause; both sides treat it alike)
    pk[k] = bad
    want = _frames_cpu(pk)
    with XMapsEngine(tb) as eng, DeviceIngest(eng, 60, activity_filter=True, capacity_events=1 << 14, max_packet_events=1 << 12, result_ring=64) as ing:
        got = []
        for p in pk:
            ing.push(p)
        ing.flush()
        got += ing.poll()
        dstat = ing.device_stats()
        assert ing.activity_sequential_packets() >= 1
    # (frames may be unsorted in time around the glitch: compare what the device kept and cut, not the depth of such a frame)
    assert len(got) == len(want) and [f.n_events for f in got] == [len(f) for f in want]
    act = IO.ActivityFilterOracle(CFG.cam_w, CFG.cam_h, int(1e6 / 60))
    assert dstat["events_appended"] == sum(len(act.process(IO.polarity_filter(p))) for p in pk)
