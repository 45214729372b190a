"""DepthReprojectionPipe with the reference's interface (python/depth_reprojection_pipe.py:37-175) and the
hot path on the GPU.

    pipe = DepthReprojectionPipe(params, stats_printer, frame_callback)
    pipe.process_events(evs)        # packets from the camera / file reader
    pipe.process_ev_frame(evs)      # one projector frame of EventCD records -> frame_callback(BGR u8)

`process_ev_frame` is the function the trigger finder calls (trigger_finder.py:172).  In the reference it
runs six NumPy/Numba/OpenCV stages; here it is one C-ABI call (xm_process_frame_aos) = three HIP kernels,
and the frame handed to `frame_callback` is a fresh (H, W, 3) uint8 BGR array exactly as before.
The packet side mirrors pipe:110-119: polarity filter -> activity-noise filter -> trigger finder -- by default (round 6) as
kernels of the device ingest (RuntimeParams.device_ingest: process_events(packet) stages the packet and returns; the frames
arrive through frame_callback as the consumer's own arrays), or on the host (the opt-out, and whenever a frame event filter or a
caller-supplied activity filter is selected) with x_maps_amd.activity_filter.ActivityNoiseFilterAlgorithm (the same kernels behind one call);
Metavision's own filter is a binary of the SDK, so the rule is this build's definition (oracle/ingest_oracle.py).
Out of scope in this build: the timing watchdog.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np

from .cam_proj_calibration import CamProjMaps, load_tables_npz
from .disp_to_depth import DisparityToDepth
from .frame_event_filter import FrameEventFilterProcessor, NoFilter
from .stats import StatsPrinter
from .trigger_finder import RobustTriggerFinder
from .x_maps_disparity import XMapsDisparity


_ACT_WARNED = False


def _warn_activity_rule_unpinned() -> None:
    """Once per process: this build's activity-noise rule (an earlier event at one of the 8 neighbours within T, inclusive; the own
    pixel does not count; per-pixel maximum stamp) is its own definition -- Metavision's ActivityNoiseFilterAlgorithm
    (depth_reprojection_pipe.py:65-67 of the reference) ships as a binary and no fixture of it exists yet
    (tests/golden/g10_metavision.npz, written by tools/pin_thirdparty.py on a reference installation).  Frames may differ from the
    reference's by the events the two rules judge differently; pass `activity_filter=` (Metavision's own) for the reference's rule."""
    global _ACT_WARNED
    if _ACT_WARNED:
        return
    _ACT_WARNED = True
    import os
    import warnings
    if not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_metavision.npz")):
        warnings.warn("x_maps_amd: the activity-noise filter runs this build's own rule, unpinned against Metavision's "
                      "ActivityNoiseFilterAlgorithm (no g10_metavision.npz fixture yet: tools/pin_thirdparty.py); pass the SDK's "
                      "filter as DepthReprojectionPipe(activity_filter=...) or RuntimeParams(activity_filter=False) to leave it out",
                      RuntimeWarning, stacklevel=3)


@dataclass
class DepthReprojectionPipe:
    params: "RuntimeParams"
    stats_printer: StatsPrinter
    frame_callback: Callable

    calib_maps: CamProjMaps = field(init=False)
    x_maps_disp: XMapsDisparity = field(init=False)
    disp_to_depth: DisparityToDepth = field(init=False)
    trigger_finder: RobustTriggerFinder = field(init=False)
    # host path: callable(pos_events) -> kept events.  None = this build's filter when RuntimeParams.activity_filter is on
    # (the default, as in the reference); plug Metavision's own filter in here where the SDK is installed
    activity_filter: Optional[Callable[[np.ndarray], np.ndarray]] = None
    fused: bool = True  # False = run the reference's six stages one by one (each still a HIP kernel)

    def __post_init__(self):
        p = self.params
        tables = getattr(p, "tables", None)
        if tables is None:
            if isinstance(p.calib, str) and p.calib.endswith(".npz"):
                tables = load_tables_npz(p.calib)  # exported from a reference installation
            elif isinstance(p.calib, str):
                # the reference's own calibration YAML: tables from the cv2-free builder (x_maps_amd/calibration.py; the
                # rectifying rotations are pinned to OpenCV's, focal length / principal point are unpinned -- DESIGN.md),
                # X-map built on the GPU.  Mirrors pipe:69-90.
                from . import calibration as calib
                cp = calib.CamProjCalibrationParams.from_yaml(p.calib, p.camera_width, p.camera_height,
                                                              p.projector_width, p.projector_height)
                tm = np.load(p.projector_time_map) if p.projector_time_map else None
                tables = calib.build_tables(cp, z_near=p.z_near, z_far=p.z_far, device=getattr(p, "device", 0),
                                            projector_time_map_rectified=tm)
                self.stats_printer.log("tables built by the cv2-free builder")
            else:
                raise ValueError("RuntimeParams.calib must be a calibration .yaml, an exported tables .npz, or set .tables")
        tables = dict(tables)
        tables.setdefault("z_near", p.z_near)
        tables.setdefault("z_far", p.z_far)
        tables["z_near"], tables["z_far"] = p.z_near, p.z_far
        cam_h, cam_w = np.asarray(tables["cam_mapx_i16"]).shape
        if (cam_w, cam_h) != (p.camera_width, p.camera_height):
            raise ValueError(f"tables are for a {cam_w}x{cam_h} camera, params say {p.camera_width}x{p.camera_height}")
        # frames cut out of the camera stream by the trigger finder are time-sorted (NoFilter): let the GPU take
        # (tmin, tmax) = (t[0], t[-1]) and verify it; an unsorted frame handed to process_ev_frame directly is detected
        # on the device and redone on the general path inside the same call, so results are exact either way
        self.calib_maps = CamProjMaps(tables, camera_perspective=p.camera_perspective,
                                      device=getattr(p, "device", 0), assume_time_sorted=True)
        self.x_maps_disp = XMapsDisparity(self.calib_maps)
        self.disp_to_depth = DisparityToDepth(stats=self.stats_printer, calib_maps=self.calib_maps,
                                              z_near=p.z_near, z_far=p.z_far)
        self.ev_filter_proc = FrameEventFilterProcessor(self.calib_maps.engine)
        self.trigger_finder = RobustTriggerFinder(projector_fps=p.projector_fps, stats=self.stats_printer,
                                                  frame_callback=self.process_ev_frame)
        # device-side ingest (row N2): raw packets go to the GPU, which filters, buffers, cuts frames and runs the hot path on
        # them without the event stream (or any index into it) coming back to the host
        self.ingest = None
        self._raw_dev, self._raw_host = {}, {}  # EVT 3.0 / 2.0 decoders (process_evt3_words / process_evt2_words), created on first use
        self._own_act_filter = None
        self._host_chain_active = False
        if self.activity_filter is None and getattr(p, "activity_filter", True):
            _warn_activity_rule_unpinned()
        # a caller-supplied activity filter (Metavision's own, where the SDK is installed) runs on the host: so does the chain
        if getattr(p, "device_ingest", True) and self.activity_filter is None:
            from .ingest import DeviceIngest
            self.ingest = DeviceIngest(self.calib_maps.engine, p.projector_fps, use_polarity=True,
                                       activity_filter=bool(getattr(p, "activity_filter", True)), want_depth=False,
                                       activity_thresh_us=self._activity_thresh_us(), activity_include_self=bool(getattr(p, "activity_include_self", False)),
                                       result_ring=int(getattr(p, "ingest_result_ring", 16)),
                                       lossless=not p.should_drop_frames)  # no_frame_dropping (the default): never lap the ring
            self._ingest_views = bool(getattr(p, "ingest_frame_views", False))
        else:
            self._ensure_host_chain()

    def _activity_thresh_us(self) -> int:
        """int(1e6 / fps) as the reference passes it (pipe:65-68); one less for the strict comparison (integer stamps)"""
        p = self.params
        return int(1e6 / p.projector_fps) - (1 if getattr(p, "activity_strict", False) else 0)

    def _ensure_host_chain(self):
        """the host chain's activity filter (pipe:65-67), made when the chain is first needed"""
        p = self.params
        if self.activity_filter is None and getattr(p, "activity_filter", True):
            from .activity_filter import ActivityNoiseFilterAlgorithm
            self._own_act_filter = ActivityNoiseFilterAlgorithm(self.calib_maps.engine, self._activity_thresh_us(),  # pipe:65-67
                                                                include_self=bool(getattr(p, "activity_include_self", False)))
            self.activity_filter = self._own_act_filter

    def _use_ingest(self) -> bool:
        """Packets go to the device ingest unless a frame event filter is selected (pipe:131-139: those filters re-order the cut
        frame's events on the host, between the trigger finder and the hot path) -- then the host chain takes the stream, from a
        clean start on either side (a switch is a user pressing E: the frames around it are not comparable anyway)."""
        if self.ingest is None:
            return False
        want_host = not isinstance(self.ev_filter_proc.selected_filter(), NoFilter)
        if want_host != self._host_chain_active:
            self._host_chain_active = want_host
            if want_host:
                self.ingest.flush()
                self._deliver_ingest_frames()
                self.ingest.reset()
                self._ensure_host_chain()
                if self._own_act_filter is not None:
                    self._own_act_filter.reset()
            self.trigger_finder.reset()
        return not want_host

    # ---- packets -> frames (host side, in front of the hot path) ---------------------------------------
    def _deliver_ingest_frames(self):
        for fr in self.ingest.poll(copy=not self._ingest_views):
            if fr.lost:  # the result ring was lapped before the host polled: the frame's images are gone (never shown)
                self.stats_printer.count("frame lost")
                continue
            self.stats_printer.count("trig ✅")
            self.stats_printer.add_metric("frame len [ms]", (fr.t_last - fr.t_first) / 1000)
            self.last_ingest_frame = fr
            # a fresh array copied out of the pinned result ring, or (RuntimeParams.ingest_frame_views) a view into it that stays
            # valid until ingest_result_ring - 1 further frames have been produced
            self.frame_callback(fr.bgr)

    def flush(self):
        if self.ingest is not None:
            self.ingest.flush()
            self._deliver_ingest_frames()

    def process_evt3_words(self, words):
        """A chunk of a recording's EVT 3.0 words (x_maps_amd.evt3.read_raw_words) instead of an EventCD packet: decoded on the
        device in front of the ingest when `device_ingest` is on (the words cross PCIe as the file stores them), on the host
        otherwise.  What Metavision's reader + process_events do together in the reference (bias_events_iterator.py:53-96)."""
        self._process_raw_words(words, 3)

    def process_evt2_words(self, words):
        """the same for EVT 2.0 words (x_maps_amd.evt2.read_raw_words): 32-bit words, the encoding of older sensors"""
        self._process_raw_words(words, 2)

    def _process_raw_words(self, words, fmt):
        from . import evt2, evt3
        mod, dt = (evt2, "<u4") if fmt == 2 else (evt3, "<u2")
        if self._use_ingest():
            dev = self._raw_dev.get(fmt)
            if dev is None:
                cls = evt2.DeviceEvt2Decoder if fmt == 2 else evt3.DeviceEvt3Decoder
                dev = self._raw_dev[fmt] = cls(self.calib_maps.engine, max_words=1 << 20,
                                               wait_for_time_base=bool(getattr(self.params, "raw_wait_for_time_base", False)))
            from ._native import XMapsTooMany
            w = np.ascontiguousarray(words, dtype=dt)

            def push(piece):
                try:
                    dev.push(self.ingest, piece)
                except XMapsTooMany:  # (vector words: up to 12 events each -- more than a packet slot holds; nothing has advanced)
                    if len(piece) < 2:
                        raise
                    push(piece[:len(piece) // 2])
                    push(piece[len(piece) // 2:])
                    return
                self._deliver_ingest_frames()
            for a in range(0, len(w), dev.max_words):
                push(w[a:a + dev.max_words])
            return
        host = self._raw_host.get(fmt)
        if host is None:
            wait = bool(getattr(self.params, "raw_wait_for_time_base", False))
            host = self._raw_host[fmt] = mod.Evt2Decoder(wait) if fmt == 2 else mod.Evt3Decoder(wait)
        evs = host.decode(words)
        if len(evs):
            self.process_events(evs)

    def process_events(self, evs):
        if self._use_ingest():
            self.ingest.push(evs)
            self._deliver_ingest_frames()
            return
        pos = evs[evs["p"] == 1]  # PolarityFilterAlgorithm(1), pipe:43,114
        if self.activity_filter is not None:
            pos = self.activity_filter(pos)
        self.trigger_finder.process_events(pos)

    # ---- one frame of events -> BGR frame (the hot path) ------------------------------------------------
    def process_ev_frame(self, evs):
        if len(evs) == 0:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")  # xmd:12
        if not isinstance(self.ev_filter_proc.selected_filter(), NoFilter):  # pipe:131-139
            with self.stats_printer.measure_time("frame ev filter"):
                n_before = len(evs)
                xr, _ = self.calib_maps.rectify_cam_coords_i16(evs)
                evs = self.ev_filter_proc.filter_events(evs, xr)
                self.stats_printer.add_metric("frame evs filtered out [%]", 100 - len(evs) / n_before * 100)
        if not self.fused:
            return self._process_ev_frame_staged(evs)
        with self.stats_printer.measure_time("x-maps frame (fused)"):
            _, bgr, st = self.calib_maps.engine.process_events(evs, use_polarity=False, want_depth=False)
        self.stats_printer.add_metric("frame evs filtered out [%]", 0.0)
        self.last_stats = st
        self.frame_callback(bgr)

    def process_ev_frames(self, frames):
        """Offline replay: a list of event frames (what the trigger finder would hand to process_ev_frame one by one) through
        the engine's multi-frame launches, `frame_callback(bgr)` once per frame, in order.  Same frames as calling
        process_ev_frame on each (no frame event filter selected); the kernels of a group keep the chip full, which a single
        frame's launches do not (groups take the column-tile K1)."""
        if not self.fused or not isinstance(self.ev_filter_proc.selected_filter(), NoFilter):
            for evs in frames:
                self.process_ev_frame(evs)
            return
        if getattr(self, "_replay_engine", None) is None:
            # an engine of its own with the library's default flags (verified shortcut with automatic redo: the mode in which
            # groups take the column tiles) and enough slots for a group; the per-frame engine declares its frames sorted
            from .engine import XMapsEngine
            self._replay_engine = XMapsEngine(self.calib_maps.tables, camera_perspective=self.params.camera_perspective,
                                              device=getattr(self.params, "device", 0), n_slots=self.replay_group)
        with self.stats_printer.measure_time("x-maps frames (fused, groups)"):
            outs = self._replay_engine.process_event_frames(list(frames), want_depth=False)
        for _, bgr in outs:
            self.frame_callback(bgr)

    def depth_frame(self, evs):
        """Same frame as process_ev_frame but returning the f32 depth map (A5 output) instead of calling back."""
        depth, _, st = self.calib_maps.engine.process_events(evs, want_bgr=False)
        self.last_stats = st
        return depth

    def _process_ev_frame_staged(self, evs):
        sp = self.stats_printer
        with sp.measure_time("ev rect"):
            xr, yr = self.calib_maps.rectify_cam_coords_i16(evs)
        with sp.measure_time("x-maps disp"):
            disp, mask = self.x_maps_disp.compute_event_disparity(events=evs, ev_x_rect_i16=xr, ev_y_rect_i16=yr)
        with sp.measure_time("disp map"):
            if self.params.camera_perspective:
                disp_map = self.calib_maps.compute_disp_map_camera_view(events=evs, inlier_mask=mask,
                                                                        ev_disparity_f32=disp)
            else:
                disp_map = self.calib_maps.compute_disp_map_projector_view(
                    ev_x_rect_i16=xr, ev_y_rect_i16=yr, inlier_mask=mask, ev_disparity_f32=disp)
        if not self.params.camera_perspective:
            disp_map = self.disp_to_depth.remap_rectified_disp_map_to_proj(disp_map)
        with sp.measure_time("disp2rgb"):
            depth_map = self.disp_to_depth.colorize_depth_from_disp(disp_map)
        self.frame_callback(depth_map)

    def select_next_frame_event_filter(self):
        new_filter = self.ev_filter_proc.select_next_filter()
        self.stats_printer.log(f"Selected event filter: {new_filter}")

    def reset(self):
        self.trigger_finder.reset()
        if self.ingest is not None:
            self.ingest.reset()  # (buffered events and the activity filter's history: the stream starts over)
        if self._own_act_filter is not None:
            self._own_act_filter.reset()

    replay_group = 16  # frames per group of process_ev_frames

    def close(self):
        if getattr(self, "_replay_engine", None) is not None:
            self._replay_engine.close()
            self._replay_engine = None
        for dev in getattr(self, "_raw_dev", {}).values():  # (before the engine they belong to)
            dev.close()
        self._raw_dev = {}
        if self.ingest is not None:
            self.ingest.close()
        if self._own_act_filter is not None:
            self._own_act_filter.close()
            self._own_act_filter = None
        self.calib_maps.engine.close()
