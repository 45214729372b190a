"""Device-side ingest ("next" row N2): raw camera packets in, finished frames out, the event stream stays in HBM.

    ing = DeviceIngest(engine, projector_fps=60, activity_filter=True)
    for packet in camera:                    # EventCD records, any polarity
        ing.push(packet)                     # H2D + a fixed sequence of launches, asynchronous
        for frame in ing.poll():             # frames finished since the last call
            show(frame.bgr)                  # fresh NumPy arrays (copied out of the pinned ring)
    ing.flush(); frames = ing.poll()

Replaces the host side of python/depth_reprojection_pipe.py:110-119 (PolarityFilterAlgorithm, ActivityNoiseFilterAlgorithm) and
python/trigger_finder.py:128-189 (RobustTriggerFinder) with kernels (x_maps_amd/csrc/xmaps_ingest.hpp); the frame that is cut
is handed to the fused K0 -> K1 -> K2 launches through a descriptor in device memory.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N
from .synthetic import EVENT_CD_DTYPE


@dataclass
class IngestFrame:
    seq: int
    n_events: int
    t_first: int
    t_last: int
    n_inliers: int
    n_index_errors: int
    live_after: int
    overflow: int
    lost: bool
    depth: np.ndarray | None
    bgr: np.ndarray | None
    push_seq: int = 0  # number (from 1) of the push whose packet cut the frame
    push_to_publish_us: float = 0.0  # the library's own clock: that push call entered -> the frame's sequence number published


class _OwnedBuffer:
    """One result buffer that left the ingest with its frame (xm_ingest_poll_owned): the base object of the NumPy array handed
    to the consumer; when the last array / view over it is gone the buffer goes back to the library's pool of pinned buffers."""
    __slots__ = ("_release", "_pool", "_ptr", "_kind", "__array_interface__")

    def __init__(self, release, pool, ptr, kind, shape, typestr):
        self._release, self._pool, self._ptr, self._kind = release, pool, ptr, kind
        self.__array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": typestr, "version": 3}

    def __del__(self):
        try:
            self._release(self._pool, self._ptr, self._kind)
        except Exception:  # (interpreter shutdown)
            pass


class DeviceIngest:
    def __init__(self, engine, projector_fps: int, use_polarity: bool = True, activity_filter: bool = False,
                 activity_thresh_us: int = 0, capacity_events: int = 0, max_packet_events: int = 0, result_ring: int = 8,
                 expected_events_per_frame: int = 0, want_depth: bool = True, want_bgr: bool = True, min_events_per_frame: int = 0,
                 launch_thread: bool = True, lossless: bool = False, activity_include_self: bool = False):
        self._e = engine
        self._lib = engine._lib
        cfg = N.xm_ingest_config()
        cfg.struct_size = C.sizeof(N.xm_ingest_config)
        cfg.projector_fps = int(projector_fps)
        cfg.use_polarity = int(use_polarity)
        cfg.activity_filter = int(activity_filter)
        cfg.activity_thresh_us = int(activity_thresh_us)
        cfg.pause_thresh_us = 0
        cfg.min_events_per_frame = int(min_events_per_frame)  # 0 = the reference's 1000 (trigger_finder.py:8)
        cfg.result_ring = int(result_ring)
        cfg.capacity_events = int(capacity_events)
        cfg.max_packet_events = int(max_packet_events)
        cfg.expected_events_per_frame = int(expected_events_per_frame)
        cfg.want_depth, cfg.want_bgr = int(want_depth), int(want_bgr)
        cfg.flags = 0 if launch_thread else N.XM_INGEST_NO_LAUNCH_THREAD  # (default: push() posts to the ingest's launch thread)
        if activity_include_self:  # (a variant of this build's activity rule: an earlier event at the event's own pixel qualifies too)
            cfg.flags |= N.XM_INGEST_ACT_SELF
        self._g = C.c_void_p(None)
        N.check(self._lib.xm_ingest_create(engine._h, C.byref(cfg), C.byref(self._g)))
        self.max_packet = int(max_packet_events) or (1 << 19)
        self.shape = (engine.out_h, engine.out_w)
        self._views = {}
        # lossless: a caller that polls after every push never loses a frame to the result ring being lapped -- at most one frame
        # is cut per push, so after result_ring - 1 pushes without a synchronisation the next push waits for the GPU first (the
        # frames cut so far are then published and the caller's poll behind this push picks them up).  Off: the reference's own
        # behaviour under load -- frames the host did not fetch in time are dropped and reported (`lost`).
        self._lossless = bool(lossless)
        self._ring = int(result_ring) if int(result_ring) > 0 else 8
        self._fr = N.xm_ingest_frame()
        self._fr_ref = C.byref(self._fr)
        self._backlog = C.c_uint64(0)
        self._pool = C.c_void_p(None)   # the pool of the frame polled last (NULL unless its buffers left the ring with it)
        self._pool_ref = C.byref(self._pool)
        self._pool_seen = None          # ... of this ingest, once a frame has left with its buffers

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            self._lib.xm_ingest_destroy(self._g)
            self._g = C.c_void_p(None)

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

    def push(self, evs: np.ndarray):
        """One packet of EventCD records; larger packets are split (every piece is a packet of its own for the trigger
        finder's once-per-packet decision, like feeding the reference smaller packets)."""
        if evs.dtype != EVENT_CD_DTYPE:
            evs = evs.astype(EVENT_CD_DTYPE)
        evs = np.ascontiguousarray(evs)
        self._backpressure(max(1, -(-len(evs) // self.max_packet)))
        if len(evs) == 0:
            N.check(self._lib.xm_ingest_push(self._g, None, 0))
            return
        for a in range(0, len(evs), self.max_packet):
            part = evs[a:a + self.max_packet]
            N.check(self._lib.xm_ingest_push(self._g, C.c_void_p(part.ctypes.data), len(part)))

    def push_pinned(self, evs: np.ndarray):
        """A packet that already lives in pinned host memory (XMapsEngine.host_empty): no staging copy."""
        assert evs.dtype == EVENT_CD_DTYPE and evs.flags.c_contiguous
        self._backpressure(max(1, -(-len(evs) // self.max_packet)))
        for a in range(0, len(evs), self.max_packet):
            part = evs[a:a + self.max_packet]
            N.check(self._lib.xm_ingest_push_pinned(self._g, C.c_void_p(part.ctypes.data), len(part)))

    def _backpressure(self, n_pushes):
        """lossless: frames not polled yet + packets whose verdict is still out (each may cut one) stay below the result ring's
        size.  Waits for verdicts only (xm_ingest_backlog) -- nothing is synchronised, the GPU's pipeline stays full.  What
        waiting cannot settle are frames the caller has not polled: the contract is a caller that polls after every push."""
        if not self._lossless:
            return
        N.check(self._lib.xm_ingest_backlog(self._g, max(1, self._ring - n_pushes), C.byref(self._backlog)))

    def _view(self, ptr, shape, ctype):
        """NumPy view of one buffer of the pinned result ring (built once per buffer, from the address: np.ctypeslib.as_array on
        a pointer costs ~0.2 ms for a 2 M-pixel frame)."""
        v = self._views.get(ptr)
        if v is None:
            n = int(np.prod(shape))
            v = np.frombuffer((ctype * n).from_address(ptr), dtype=np.dtype(ctype)).reshape(shape)
            self._views[ptr] = v
        return v

    def poll(self, copy: bool = True) -> list[IngestFrame]:
        """Frames finished since the last call.  copy=True (default): depth / bgr are the consumer's own arrays, as the reference's
        frame_callback gets them -- no lifetime rule; they are the pinned buffers the frame's DMA filled, taken out of the result
        ring (xm_ingest_poll_owned: nothing is copied on the host; a buffer returns to the library's pool when the last array
        over it is dropped; only when the pool is exhausted is the frame copied out of the ring instead).  copy=False: views into
        the pinned result ring, valid until `result_ring` - 1 further frames have been cut (the lifetime xm_ingest_frame
        documents)."""
        out = []
        fr = self._fr
        h, w = self.shape
        lib = self._lib
        while True:
            rc = lib.xm_ingest_poll_owned(self._g, self._fr_ref, self._pool_ref) if copy else lib.xm_ingest_poll(self._g, self._fr_ref)
            if rc < 0:
                N.check(rc)
            if rc == 0:
                break
            depth = bgr = None
            lost = bool(fr.lost)
            if copy and fr.owned:
                pool, rel = self._pool.value, lib.xm_frame_pool_release
                self._pool_seen = pool
                if fr.depth:
                    depth = np.asarray(_OwnedBuffer(rel, pool, fr.depth, 0, (h, w), "<f4"))
                if fr.bgr:
                    bgr = np.asarray(_OwnedBuffer(rel, pool, fr.bgr, 1, (h, w, 3), "|u1"))
            else:
                if fr.depth:
                    depth = self._view(fr.depth, (h, w), C.c_float)
                    depth = depth.copy() if copy else depth
                if fr.bgr:
                    bgr = self._view(fr.bgr, (h, w, 3), C.c_uint8)
                    bgr = bgr.copy() if copy else bgr
                if copy and not lost and (depth is not None or bgr is not None) and not lib.xm_ingest_frame_valid(self._g, fr.seq):
                    lost, depth, bgr = True, None, None  # the ring was lapped while the frame was being copied out: a torn copy is no frame
            out.append(IngestFrame(int(fr.seq), int(fr.n_events), int(fr.t_first), int(fr.t_last), int(fr.n_inliers),
                                   int(fr.n_index_errors), int(fr.live_after), int(fr.overflow), lost, depth, bgr, int(fr.push_seq), float(fr.push_to_publish_us)))
        return out

    def pool_stats(self) -> dict:
        """The pool of pinned result buffers behind poll(copy=True): buffers made so far (beyond the result ring's own), in
        consumers' hands, spare."""
        if not self._pool_seen or not self._g.value:  # (the pool lives as long as the ingest, or a consumer's buffer: ask while the ingest is open)
            return {"allocated": 0, "outstanding": 0, "spare": 0}
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        N.check(self._lib.xm_frame_pool_stats(C.c_void_p(self._pool_seen), C.byref(a), C.byref(b), C.byref(c)))
        return {"allocated": int(a.value), "outstanding": int(b.value), "spare": int(c.value)}

    def device_stats(self) -> dict:
        """The device's counters once everything pushed so far has run (synchronises): frames cut, events appended behind the
        filters, events dropped because the ring had no room, events still buffered."""
        a, b, c, d = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        N.check(self._lib.xm_ingest_device_stats(self._g, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"frames_cut": int(a.value), "events_appended": int(b.value), "events_dropped": int(c.value), "events_live": int(d.value)}

    def activity_sequential_packets(self) -> int:
        """activity filter: packets so far that the device judged sequentially (synchronises)"""
        n = C.c_uint64(0)
        N.check(self._lib.xm_ingest_activity_stats(self._g, C.byref(n)))
        return int(n.value)

    def activity_fused_first_passes(self) -> int:
        """activity filter: packets so far whose first pass rode on the packet before's counting launch (synchronises)"""
        n = C.c_uint64(0)
        N.check(self._lib.xm_ingest_fused_first_passes(self._g, C.byref(n)))
        return int(n.value)

    def host_stats(self) -> dict:
        """What the calling thread has paid inside push() so far (xm_ingest_host_stats)."""
        n, sec, waits, wsec = C.c_uint64(0), C.c_double(0.0), C.c_uint64(0), C.c_double(0.0)
        N.check(self._lib.xm_ingest_host_stats(self._g, C.byref(n), C.byref(sec), C.byref(waits), C.byref(wsec)))
        work = float(sec.value) - float(wsec.value)
        return {"pushes": int(n.value), "host_seconds_in_push": float(sec.value), "staging_waits": int(waits.value),
                "seconds_waiting_for_the_gpu": float(wsec.value),
                "us_per_push": (float(sec.value) / n.value * 1e6) if n.value else 0.0,
                "us_per_push_without_waits": (work / n.value * 1e6) if n.value else 0.0}

    def flush(self):
        N.check(self._lib.xm_ingest_flush(self._g))

    def reset(self):
        N.check(self._lib.xm_ingest_reset(self._g))
