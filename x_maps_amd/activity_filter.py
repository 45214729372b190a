"""The activity-noise filter alone: what `self.act_filter.process_events(self.pos_events_buf, act_out_buf)` is in the reference's
pipe (python/depth_reprojection_pipe.py:65-67 builds `ActivityNoiseFilterAlgorithm(width, height, int(1e6 / fps))`, :116-117
runs it on every packet behind the polarity filter), for a host that keeps the trigger finder on the CPU.

    act = ActivityNoiseFilterAlgorithm(engine, int(1e6 / fps))
    kept = act.process_events(pos_events)          # EventCD records in, the kept ones out (a fresh array, stream order)

Metavision's filter comes as a binary with the SDK, so the rule is this build's own definition (xmaps_ingest.hpp; the sequential
restatement and what is known of the differences: oracle/ingest_oracle.py): an event is kept iff an EARLIER event of the stream
at one of its 8 neighbouring pixels has t - t' <= threshold; every event then joins its pixel's history.  The rule runs on the
GPU (xm_activity_process: the same kernels as the device ingest's filter); the history stays there between calls.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .synthetic import EVENT_CD_DTYPE


class ActivityNoiseFilterAlgorithm:
    def __init__(self, engine, threshold_us: int, max_packet_events: int = 0, include_self: bool = False):
        self._lib = engine._lib
        self._f = C.c_void_p(None)
        self.threshold_us = int(threshold_us)
        N.check(self._lib.xm_activity_create(engine._h, self.threshold_us, int(max_packet_events), C.byref(self._f)))
        if include_self:  # (a variant of the rule: the event's own pixel counts; the strict comparison is threshold_us - 1)
            N.check(self._lib.xm_activity_set_rule(self._f, 1))

    def process_events(self, evs: np.ndarray, return_mask: bool = False):
        """evs: EventCD records (every one of them takes part: hand in the polarity filter's output) -> the kept records."""
        if evs.dtype != EVENT_CD_DTYPE:
            evs = evs.astype(EVENT_CD_DTYPE)
        evs = np.ascontiguousarray(evs)
        keep = np.empty(len(evs), np.uint8)
        if len(evs):
            N.check(self._lib.xm_activity_process(self._f, C.c_void_p(evs.ctypes.data), len(evs), C.c_void_p(keep.ctypes.data), None))
        mask = keep.view(bool)
        return mask if return_mask else evs[mask]

    __call__ = process_events

    def sequential_packets(self) -> int:
        """packets so far that the device judged sequentially (stamps running backwards, > 8 thresholds in one packet)"""
        n = C.c_uint64(0)
        N.check(self._lib.xm_activity_stats(self._f, C.byref(n)))
        return int(n.value)

    def reset(self):
        N.check(self._lib.xm_activity_reset(self._f))

    def close(self):
        if getattr(self, "_f", None) is not None and self._f.value:
            self._lib.xm_activity_destroy(self._f)
            self._f = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
