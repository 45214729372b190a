"""CPU restatement of the ingest chain in front of the hot path -- TEST INFRASTRUCTURE ONLY (see xmaps_oracle.py).

  polarity_filter        PolarityFilterAlgorithm(1) as the reference uses it (python/depth_reprojection_pipe.py:43,114)
  activity_filter        OWN DEFINITION (Metavision's ActivityNoiseFilterAlgorithm is installed as a binary with the SDK,
                         SURVEY.md 8(c); parity unpinned): an event is kept iff an EARLIER event of the (positive) stream at one
                         of its 8 neighbouring pixels has t - t' <= thresh (thresh = int(1e6 / fps), pipe:65-68); every event,
                         kept or not, then joins its pixel's history (the pixel's LARGEST stamp so far).  Sequential by
                         construction; the device evaluates the same rule in parallel (xmaps_ingest.hpp).
                         Beside the published OpenEB header (sdk/modules/cv/.../activity_noise_filter_algorithm{,_impl}.h) AS
                         REMEMBERED -- not readable offline, so every line here is to be checked by tools/pin_thirdparty.py
                         where the SDK exists: it keeps one `last_ts` per pixel, stores the event's stamp there, and keeps
                         the event iff a pixel of the 3 x 3 neighbourhood was written more recently than `t - threshold`.
                         Known or possible differences, each a one-line change here and in act_keep():
                           (1) comparison: this rule is `t - t' <= T`; a strict `t' > t - T` is thresh = T - 1 (integers);
                           (2) history: this rule keeps the pixel's MAXIMUM stamp, OpenEB overwrites with the LATEST event's
                               -- identical on a time-ordered stream, which a camera's is;
                           (3) the event's own pixel: excluded here (an isolated hot pixel firing repeatedly is noise); if
                               OpenEB's window includes the centre, events repeating at one pixel within T are kept there
                               (include_self=True here, XM_INGEST_ACT_SELF / RuntimeParams.activity_include_self there);
                           (4) borders: the neighbourhood is clipped to the sensor here; coordinates outside the sensor are
                               dropped (NumPy would raise), OpenEB's behaviour there is not known.
  TriggerFinderOracle    find_trigger / process_events of python/trigger_finder.py:128-189 written out over plain arrays
                         (pinned by tests/golden/g5_trigger.npz in tests/test_oracle_ingest.py)
"""
from __future__ import annotations

import numpy as np


_EVENT_CD = np.dtype({"names": ["x", "y", "p", "t"], "formats": ["<u2", "<u2", "<i2", "<i8"], "offsets": [0, 2, 4, 8], "itemsize": 16})


def polarity_filter(evs):
    return evs[evs["p"] == 1]


class ActivityFilterOracle:
    def __init__(self, width, height, thresh_us, include_self=False):
        self.w, self.h, self.thresh, self.include_self = int(width), int(height), int(thresh_us), bool(include_self)
        self.last = np.full((self.h, self.w), np.iinfo(np.int64).min, np.int64)
        self.has = np.zeros((self.h, self.w), bool)

    def process(self, evs):
        """evs: positive events in stream order -> the kept ones."""
        keep = np.zeros(len(evs), bool)
        xs, ys, ts = evs["x"].astype(np.int64), evs["y"].astype(np.int64), evs["t"].astype(np.int64)
        last, has, w, h, T, own = self.last, self.has, self.w, self.h, self.thresh, self.include_self
        for i in range(len(evs)):
            x, y, t = xs[i], ys[i], ts[i]
            y0, y1, x0, x1 = max(y - 1, 0), min(y + 1, h - 1), max(x - 1, 0), min(x + 1, w - 1)
            k = False
            for yy in range(y0, y1 + 1):
                for xx in range(x0, x1 + 1):
                    if (own or yy != y or xx != x) and has[yy, xx] and t - last[yy, xx] <= T:  # (own: variant (3) above)
                        k = True
            keep[i] = k
            if not has[y, x] or t > last[y, x]:
                last[y, x] = t
            has[y, x] = True
        return evs[keep]


def activity_filter_c(lib, evs, state, thresh_us, include_self=False):
    """The same rule through oracle/xmaps_oracle.c:xmo_activity_filter (fast; pinned to ActivityFilterOracle in
    tests/test_oracle_ingest.py).  state = (last int64[h, w], has uint8[h, w]) carried by the caller."""
    import ctypes as C
    last, has = state
    rec = np.zeros(len(evs), _EVENT_CD)  # (np.concatenate of structured arrays drops the record's padding: always re-pack)
    for k in ("x", "y", "p", "t"):
        rec[k] = evs[k]
    keep = np.zeros(len(evs), np.uint8)
    lib.xmo_activity_filter2.restype = C.c_int64
    lib.xmo_activity_filter2(C.c_void_p(rec.ctypes.data), C.c_int64(len(rec)), C.c_int(has.shape[1]), C.c_int(has.shape[0]), C.c_int64(int(thresh_us)),
                             C.c_void_p(last.ctypes.data), C.c_void_p(has.ctypes.data), C.c_void_p(keep.ctypes.data), C.c_int(1 if include_self else 0))
    return evs[keep.view(bool)]


class ActivityFilterC:
    """ActivityFilterOracle's interface over the C form"""

    def __init__(self, width, height, thresh_us, include_self=False):
        import c_oracle
        self.lib = c_oracle.load(False)
        self.thresh, self.include_self = int(thresh_us), bool(include_self)
        self.state = (np.zeros((int(height), int(width)), np.int64), np.zeros((int(height), int(width)), np.uint8))

    def process(self, evs):
        return activity_filter_c(self.lib, evs, self.state, self.thresh, self.include_self)


class TriggerFinderOracle:
    """python/trigger_finder.py:91-189 without the buffer-pool plumbing (no frame dropping: should_drop is never set when
    RuntimeParams.no_frame_dropping is True, the default of this build)."""

    def __init__(self, projector_fps, pause_thresh_us=40, min_events=1000):
        self.period = 1e6 / projector_fps
        self.pause, self.min_events = pause_thresh_us, min_events
        self.buf = None
        self.frames = []
        self.ok = self.fail = 0

    def process_events(self, evs):
        if len(evs):
            self.buf = evs if self.buf is None or len(self.buf) == 0 else np.concatenate((self.buf, evs))
        if self.buf is None or len(self.buf) == 0:
            return
        if self.buf["t"][-1] - self.buf["t"][0] < self.period:
            return
        evs, self.buf = self.buf, None
        t = evs["t"]
        pauses = np.nonzero(np.diff(t) >= self.pause)[0]
        for prev, nxt in zip(pauses[:-1], pauses[1:]):
            gap = t[nxt] - t[prev]
            if gap > self.period / 2:
                if gap <= self.period and nxt - prev > self.min_events:
                    self.frames.append(evs[prev + 2:nxt - 2].copy())
                    self.buf = evs[nxt - 2:]
                    self.ok += 1
                else:
                    self.buf = evs[nxt:]
                    self.fail += 1
                return
        self.fail += 1
