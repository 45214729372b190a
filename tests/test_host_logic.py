"""CPU: host-side logic around the hot path (frame segmentation, sharding arithmetic, synthetic rig)."""
import os

import numpy as np

from x_maps_amd import synthetic as S
from x_maps_amd.sharded import shard_bounds
from x_maps_amd.stats import StatsPrinter
from x_maps_amd.trigger_finder import RobustTriggerFinder


def test_trigger_finder_matches_reference_run(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_trigger.npz"))
    ev = np.zeros(len(g["t"]), S.EVENT_CD_DTYPE)
    ev["x"], ev["y"], ev["t"], ev["p"] = g["x"], g["y"], g["t"], 1
    frames, st = [], StatsPrinter()
    tf = RobustTriggerFinder(int(g["fps"]), lambda e: frames.append(e.copy()), st)
    cuts = g["packet_cuts"]
    for a, b in zip(cuts[:-1], cuts[1:]):
        tf.process_events(ev[a:b])
    assert len(frames) == int(g["n_frames"])
    assert [len(f) for f in frames] == list(g["frame_len"])
    assert [f["t"][0] for f in frames] == list(g["frame_first_t"])
    assert [f["t"][-1] for f in frames] == list(g["frame_last_t"])
    assert st.counters["trig ✅"] == int(g["trig_ok"]) and st.counters["trig ❌"] == int(g["trig_fail"])


def test_trigger_finder_drop_and_reset():
    frames = []
    tf = RobustTriggerFinder(60, frames.append)
    ev = S.make_events(S.C_TINY, n=2000)
    tf.drop_frame()
    tf.process_events(ev[:0])  # nothing buffered: the drop request stays pending (trigger_finder.py:125-129)
    assert tf.should_drop
    tf.process_events(ev[:1000])  # whole packets starting inside the first frame period are dropped
    assert not tf.should_drop and not tf._chunks
    tf.drop_frame()
    tf.reset()
    assert not tf.should_drop and not tf._chunks and not frames


def test_shard_bounds_partition_exactly():
    for n in (0, 1, 7, 1_000_000, 10_000_001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


def test_synthetic_rig_is_deterministic_and_shaped_like_the_configs():
    a, b = S.make_events(S.C_TINY, frame=2), S.make_events(S.C_TINY, frame=2)
    assert a.dtype.itemsize == 16 and np.array_equal(a, b)
    assert (np.diff(a["t"]) >= 0).all() and a["x"].max() < S.C_TINY.cam_w and a["y"].max() < S.C_TINY.cam_h
    tb = S.make_tables(S.C_1M)
    assert (tb["rect_w"], tb["rect_h"]) == (1760, 1320) and tb["proj_x_map"].shape == (1320, 640)
    assert tb["disp_proj_mapxy_i16"].shape == (480, 640, 2) and (tb["proj_x_map"][:, 0] == 0).all()
    tb10 = S.make_tables(S.C_10M)
    assert (tb10["rect_w"], tb10["rect_h"]) == (3520, 1980) and tb10["proj_x_map"].shape == (1980, 1280)


def test_keyboard_cb_honours_metavisions_key_release_contract():
    """processor.py:96-105: the window calls keyboard_cb(key, scancode, action, mods) on press, repeat AND release; the reference
    acts on UIAction.RELEASE only and compares UIKeyEvent members.  Enum members (any object with a .name), GLFW integers and
    plain strings are understood; Metavision itself is not importable here."""
    import enum

    from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor, FakeWindow

    class UIAction(enum.Enum):  # (the SDK's enums, reduced to what the callback looks at)
        RELEASE = 0
        PRESS = 1
        REPEAT = 2

    class UIKeyEvent(enum.Enum):
        KEY_ESCAPE = 256
        KEY_Q = 81
        KEY_E = 69
        KEY_S = 83
        KEY_A = 65

    class Pipe:
        filters = 0

        def select_next_frame_event_filter(self):
            self.filters += 1

    proc = DepthReprojectionProcessor(params=None)
    proc._pipe, proc._window = Pipe(), FakeWindow()
    for action in (UIAction.PRESS, UIAction.REPEAT):
        for key in UIKeyEvent:
            proc.keyboard_cb(key, 0, action, 0)
    assert proc._pipe.filters == 0 and not proc._window.should_close() and not proc.stats_printer.silent
    proc.keyboard_cb(UIKeyEvent.KEY_A, 0, UIAction.RELEASE, 0)
    assert proc._pipe.filters == 0 and not proc._window.should_close()
    proc.keyboard_cb(UIKeyEvent.KEY_E, 0, UIAction.RELEASE, 0)  # one press-release cycle = one step
    assert proc._pipe.filters == 1
    proc.keyboard_cb(UIKeyEvent.KEY_S, 0, UIAction.RELEASE, 0)
    assert proc.stats_printer.silent
    proc.keyboard_cb(69, 0, 0, 0)  # GLFW: key 'E', action GLFW_RELEASE
    proc.keyboard_cb(69, 0, 1, 0)  # ... GLFW_PRESS
    proc.keyboard_cb("e")          # a headless driver
    assert proc._pipe.filters == 3
    proc.keyboard_cb(UIKeyEvent.KEY_Q, 0, UIAction.RELEASE, 0)
    assert proc._window.should_close()
    proc._window = FakeWindow()
    proc.keyboard_cb(UIKeyEvent.KEY_ESCAPE, 0, UIAction.RELEASE, 0)
    assert proc._window.should_close()


def test_stats_are_bounded():
    """a 60 Hz live loop adds values per frame for as long as it runs: count / mean / extrema stay exact, memory does not grow"""
    from x_maps_amd import stats
    sp = stats.StatsPrinter()
    for i in range(10 * stats.WINDOW):
        sp.add_metric("m", float(i))
        with sp.measure_time("t"):
            pass
        sp.log(f"line {i}")
    m = sp.metrics["m"]
    assert m.count == 10 * stats.WINDOW and m.min == 0.0 and m.max == 10 * stats.WINDOW - 1 and abs(m.mean() - (10 * stats.WINDOW - 1) / 2) < 1e-9
    assert len(m.recent) == stats.WINDOW == len(sp.timers["t"].recent) == len(sp.logs) and m[-1] == m.max
