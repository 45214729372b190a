"""CPU: the ingest oracle (oracle/ingest_oracle.py) against the reference's own trigger-finder run (golden G5), against the
product's host trigger finder, and the activity rule on a hand-checked example."""
import os

import numpy as np

import ingest_oracle as IO
from x_maps_amd import synthetic as S
from x_maps_amd.trigger_finder import RobustTriggerFinder


def _g5(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_trigger.npz"))
    ev = np.zeros(len(g["t"]), S.EVENT_CD_DTYPE)
    ev["x"], ev["y"], ev["t"], ev["p"] = g["x"], g["y"], g["t"], 1
    return g, ev


def test_trigger_oracle_matches_reference_run(golden_dir):
    g, ev = _g5(golden_dir)
    tf = IO.TriggerFinderOracle(int(g["fps"]))
    cuts = g["packet_cuts"]
    for a, b in zip(cuts[:-1], cuts[1:]):
        tf.process_events(ev[a:b])
    assert len(tf.frames) == int(g["n_frames"])
    assert [len(f) for f in tf.frames] == list(g["frame_len"])
    assert [f["t"][0] for f in tf.frames] == list(g["frame_first_t"])
    assert [f["t"][-1] for f in tf.frames] == list(g["frame_last_t"])
    assert tf.ok == int(g["trig_ok"]) and tf.fail == int(g["trig_fail"])


def test_trigger_oracle_equals_product_host_finder_on_a_noisy_stream():
    rng = np.random.default_rng(1)
    chunks = []
    for f in range(12):
        start = 3_000_000 + f * 16_600
        tt = np.unique(np.concatenate((np.sort(rng.integers(0, 13_000, 2600)) + start, np.arange(start, start + 13_000, 25))))
        if f % 3 == 2:
            tt = np.concatenate((tt, [start + 14_500]))  # a noise event inside the dark gap
        if f == 7:
            tt = tt[tt < start + 6_000]  # a frame that stops half way: long pause, implausible pair
        ev = np.zeros(len(tt), S.EVENT_CD_DTYPE)
        ev["t"], ev["p"] = tt, 1
        ev["x"] = rng.integers(0, 64, len(tt))
        ev["y"] = rng.integers(0, 48, len(tt))
        chunks.append(ev)
    stream = np.concatenate(chunks)
    frames = []
    tf_prod = RobustTriggerFinder(60, lambda e: frames.append(e.copy()))
    tf_or = IO.TriggerFinderOracle(60)
    packet = int(1e6 / 60 / 4)
    edges = np.arange(stream["t"][0], stream["t"][-1] + packet, packet)
    cuts = np.searchsorted(stream["t"], edges)
    for a, b in zip(cuts[:-1], cuts[1:]):
        tf_prod.process_events(stream[a:b])
        tf_or.process_events(stream[a:b])
    assert len(frames) == len(tf_or.frames) >= 4
    for a, b in zip(frames, tf_or.frames):
        assert np.array_equal(a, b)


def test_activity_rule_on_a_hand_checked_example():
    T = 100
    ev = np.zeros(7, S.EVENT_CD_DTYPE)
    #            lone      neighbour    same px      far in time   diagonal      border px     neighbour of the border px
    ev["x"] = [10,        11,          11,          10,           12,           0,            1]
    ev["y"] = [10,        10,          10,          10,           11,           0,            0]
    ev["t"] = [1000,      1050,        1060,        1300,         1390,         2000,         2100]
    ev["p"] = 1
    f = IO.ActivityFilterOracle(64, 48, T)
    kept = f.process(ev)
    # e0: nothing before it.  e1: neighbour e0 50 us earlier -> kept.  e2: same pixel as e1 does not count, e0 (x=10) 60 us
    # earlier does -> kept.  e3: neighbours (11,10) last fired at 1060: 240 us ago -> dropped.  e4: diagonal neighbour
    # (11,10) at 1060 is 330 us ago -> dropped.  e5: border pixel, nothing around.  e6: neighbour (0,0) 100 us earlier -> kept.
    assert list(kept["t"]) == [1050, 1060, 2100]
    # state carries over to the next packet: (12,11) fired at 1390
    ev2 = np.zeros(1, S.EVENT_CD_DTYPE)
    ev2["x"], ev2["y"], ev2["t"], ev2["p"] = 13, 12, 1480, 1
    assert len(f.process(ev2)) == 1


def test_activity_rule_c_form_equals_the_python_form():
    """oracle/xmaps_oracle.c:xmo_activity_filter (used on ESL-size streams) == ActivityFilterOracle, packet by packet, on sorted
    and unsorted streams"""
    rng = np.random.default_rng(3)
    for sort in (True, False):
        n = 4000
        ev = np.zeros(n, S.EVENT_CD_DTYPE)
        t = rng.integers(0, 30_000, n)
        ev["t"] = 10_000 + (np.sort(t) if sort else t)
        ev["x"], ev["y"], ev["p"] = rng.integers(0, 40, n), rng.integers(0, 30, n), 1
        for own in (False, True):  # (own: the variant whose 3 x 3 window includes the event's own pixel)
            a, b = IO.ActivityFilterOracle(40, 30, 900, include_self=own), IO.ActivityFilterC(40, 30, 900, include_self=own)
            kept = 0
            for k in range(0, n, 700):
                want, got = a.process(ev[k:k + 700]), b.process(ev[k:k + 700])
                assert np.array_equal(want, got)
                kept += len(want)
            assert 0 < len(want) < 700
            if own:
                assert kept > kept_default  # (events repeating at one pixel within T are kept by the variant only)
            kept_default = kept
