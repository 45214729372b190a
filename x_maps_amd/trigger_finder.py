"""Host-side frame segmentation in front of the hot path ("next" row N2 of SURVEY.md section 8(f)).

Behavioural restatement of the reference's RobustTriggerFinder (python/trigger_finder.py:91-189): buffer at
least one projector period of events, look for inter-event pauses >= 40 us, and hand the span between two
consecutive pauses to `frame_callback` when it is longer than half a period, not longer than one period and
holds more than 1000 events (2 events trimmed on each side).  Pinned by tests/golden/g5_trigger.npz, which
was produced by the reference's own class.  Pure NumPy on the host: O(N) diff per period, not a GPU job.
"""
from __future__ import annotations

from typing import Callable, List

import numpy as np

MIN_EVENTS_PER_FRAME = 1000  # trigger_finder.py:8


class RobustTriggerFinder:
    frame_paused_thresh_us = 40  # trigger_finder.py:98

    def __init__(self, projector_fps: int, frame_callback: Callable[[np.ndarray], None], stats=None):
        self.projector_fps = projector_fps
        self.frame_callback = frame_callback
        self.stats = stats
        self._chunks: List[np.ndarray] = []
        self.should_drop = False
        self.last_frame_start_us = -1

    # -- buffer ------------------------------------------------------------------------------------
    @property
    def frame_len_ms(self):
        return 1e3 / self.projector_fps

    def _span_us(self):
        if not self._chunks:
            return -1
        first, last = self._chunks[0]["t"][0], self._chunks[-1]["t"][-1]
        return -1 if (first < 0 or last < 0) else last - first

    def _drop_oldest(self, drop_len_ms) -> bool:
        """Drop whole buffered packets until one frame length of time is gone (trigger_finder.py:62-75)."""
        if not self._chunks:
            return False
        until = self._chunks[0]["t"][0] + drop_len_ms * 1000
        dropped = False
        while self._chunks and self._chunks[0]["t"][0] < until:
            self._chunks.pop(0)
            dropped = True
        return dropped

    def reset(self):
        self._chunks.clear()
        self.should_drop = False
        self.last_frame_start_us = -1

    def drop_frame(self):
        self.should_drop = True

    def _count(self, key):
        if self.stats is not None:
            self.stats.count(key)

    # -- per packet ----------------------------------------------------------------------------------
    def process_events(self, evs: np.ndarray):
        if len(evs):
            self._chunks.append(evs)
        if self.should_drop:
            if not self._drop_oldest(self.frame_len_ms):
                return  # not a frame's worth buffered yet
            self._count("frames dropped")
            self.should_drop = False
        if not self._chunks or self._span_us() < 1e6 / self.projector_fps:
            return
        if self.stats is not None:
            self.stats.add_metric("evs in buf", sum(len(c) for c in self._chunks))
        self._count("trig ✅" if self.find_trigger() / 1000 > 0 else "trig ❌")

    def find_trigger(self):
        evs = self._chunks[0] if len(self._chunks) == 1 else np.concatenate(self._chunks)
        self._chunks = []
        t = evs["t"]
        period = 1e6 / self.projector_fps
        pauses = np.nonzero(np.diff(t) >= self.frame_paused_thresh_us)[0]
        for prev_idx, next_idx in zip(pauses[:-1], pauses[1:]):
            gap = t[next_idx] - t[prev_idx]
            if gap <= period / 2:
                continue
            if gap <= period and next_idx - prev_idx > MIN_EVENTS_PER_FRAME:
                self.frame_callback(evs[prev_idx + 2:next_idx - 2])  # the hot path
                start, end = t[prev_idx + 2], t[next_idx - 2]
                if self.stats is not None:
                    self.stats.add_metric("frame len [ms]", (end - start) / 1000)
                    if self.last_frame_start_us != -1:
                        self.stats.add_metric("frame interval [ms]", (start - self.last_frame_start_us) / 1000)
                self.last_frame_start_us = start
                rest = evs[next_idx - 2:]
                if len(rest):
                    self._chunks.append(rest)
                return start
            rest = evs[next_idx:]  # long gap but not a plausible frame: discard up to it
            if len(rest):
                self._chunks.append(rest)
            return -1
        return -1
