"""Per-frame de-duplication filters ("next" row N3) with the reference's class names and call signature
(python/frame_event_filter.py:6-151): `filter.filter_events(events, xp_i16) -> events`.  The polarity selection
`events[events["p"] == 1]` and the map extents (`max + 1`) are host NumPy like in the reference; the per-pixel
first/last resolution and the raster-ordered compaction run on the GPU (xm_frame_event_filter).

Reference behaviour note: the reference fills its "first event" maps through reversed views
(`event_map[y[::-1], x[::-1]] = t[::-1]`, frame_event_filter.py:49,76-81,112).  NumPy (1.26 and 2.2 checked) iterates such
an assignment in memory order, i.e. exactly like the forward one, so AS IT RUNS every filter keeps the LAST event per cell.
`intended_semantics=False` (default) reproduces that -- it is what the golden vectors captured from the reference contain;
`intended_semantics=True` keeps the first event, as the class names say."""
from __future__ import annotations

from collections import deque

import numpy as np

XM_FILTER_FIRST_PER_YT, XM_FILTER_FIRST_PER_XY, XM_FILTER_LAST_PER_XY, XM_FILTER_MEAN_FIRST_LAST_PER_XY = 1, 2, 3, 4


class FrameEventFilter:
    def filter_events(self, events, xp_i16):
        raise NotImplementedError()


class NoFilter(FrameEventFilter):
    def filter_events(self, events, xp_i16):
        return events

    def __str__(self):
        return "NoFilter"


class _GpuXYFilter(FrameEventFilter):
    filter_id = 0

    def __init__(self, engine=None, intended_semantics: bool = False):
        self.engine = engine
        self.intended_semantics = intended_semantics

    def filter_events(self, events, xp_i16):
        events = events[events["p"] == 1]
        if len(events) == 0:
            raise ValueError("zero-size array to reduction operation maximum which has no identity")  # events["y"].max()
        shape = (int(events["y"].max()) + 1, int(events["x"].max()) + 1)
        return self.engine.frame_event_filter(self.filter_id, events, None, shape, self.intended_semantics)

    def __str__(self):
        return type(self).__name__


class LastEventPerXYFilter(_GpuXYFilter):
    filter_id = XM_FILTER_LAST_PER_XY


class FirstEventPerXYFilter(_GpuXYFilter):
    filter_id = XM_FILTER_FIRST_PER_XY


class MeanFirstLastEventPerXYFilter(_GpuXYFilter):
    filter_id = XM_FILTER_MEAN_FIRST_LAST_PER_XY


class FirstEventPerYTFilter(_GpuXYFilter):
    filter_id = XM_FILTER_FIRST_PER_YT

    def filter_events(self, events, xp_i16):
        events = events[events["p"] == 1]
        xp_i16 = np.asarray(xp_i16)
        if len(events) != len(xp_i16):  # the reference pairs the p == 1 events with xp element-wise
            raise ValueError(f"shape mismatch: {len(events)} positive events vs {len(xp_i16)} x-proj values")
        if len(events) == 0:
            raise ValueError("zero-size array to reduction operation maximum which has no identity")
        shape = (int(events["y"].max()) + 1, int(xp_i16.max()) + 1)
        return self.engine.frame_event_filter(self.filter_id, events, xp_i16, shape, self.intended_semantics)


class FrameEventFilterProcessor:
    """Cycled by the 'E' key in the reference (python/frame_event_filter.py:131-151)."""

    def __init__(self, engine=None):
        self.filters = deque((NoFilter(), FirstEventPerYTFilter(engine), FirstEventPerXYFilter(engine),
                              LastEventPerXYFilter(engine), MeanFirstLastEventPerXYFilter(engine)))

    def selected_filter(self):
        return self.filters[0]

    def filter_events(self, evs, xp_i16):
        return self.selected_filter().filter_events(evs, xp_i16)

    def select_next_filter(self):
        self.filters.rotate(-1)
        return self.selected_filter()
