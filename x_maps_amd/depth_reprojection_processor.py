"""RuntimeParams / DepthReprojectionProcessor with the reference's interface
(python/depth_reprojection_processor.py:13-36, 50-114):

    with DepthReprojectionProcessor(params) as proc:
        for evs in event_packets:
            proc.process_events(evs)
            if proc.should_close(): break

The window is pluggable; without Metavision's MTWindow the reference's own FakeWindow behaviour is used
(processor.py:39-47).  Frames reach `window.show_async(bgr)` exactly as in the reference.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional

from .depth_reprojection_pipe import DepthReprojectionPipe
from .stats import StatsPrinter


@dataclass
class RuntimeParams:
    camera_width: int
    camera_height: int

    projector_width: int
    projector_height: int

    projector_fps: int

    z_near: float
    z_far: float

    calib: Any  # path of the exported tables (.npz); the reference takes the calibration YAML here

    projector_time_map: Optional[str] = None

    no_frame_dropping: bool = True

    camera_perspective: bool = False

    # additions of this build (defaults keep the reference's positional signature valid)
    tables: Optional[dict] = None  # pre-built tables instead of `calib`
    device: int = 0
    # polarity / activity filter + buffering + frame segmentation on the GPU (x_maps_amd/ingest.py): process_events(packet) stages
    # the packet and returns, the frames are cut and processed on the device and handed to frame_callback as the consumer's own
    # arrays (the pinned buffers their DMA filled).  THE DEFAULT since round 6 -- the same frames as the host chain, which stays
    # as the opt-out (False: NumPy polarity mask + one GPU call per packet for the activity filter + the NumPy trigger finder,
    # 20 x slower) and takes over by itself while a frame event filter (key E) or a caller-supplied activity filter is selected.
    device_ingest: bool = True
    # The activity-noise filter behind the polarity filter, on every packet, as the reference runs it unconditionally
    # (depth_reprojection_pipe.py:65-67,116-117): kernels of the device ingest, or one GPU call per packet in front of the host's
    # trigger finder.  The rule is this build's own definition (Metavision's is a binary: oracle/ingest_oracle.py); False
    # switches the stage off (round 4's default).
    activity_filter: bool = True
    # the two variants of that rule a fixture of Metavision's filter (tools/pin_thirdparty.py) may call for, as configuration:
    # a strict comparison t - t' < T (= threshold T - 1 on integer stamps) and a 3 x 3 window that includes the event's own pixel
    activity_strict: bool = False
    activity_include_self: bool = False
    # device ingest only: hand frame_callback / window.show_async a VIEW into the ingest's ring of pinned result buffers instead of
    # an array of the consumer's own.  LIFETIME of such a view: until `ingest_result_ring` - 1 further frames have been produced.
    # (Since round 6 the default frames cost no host copy either -- xm_ingest_poll_owned --, so views only save the pool.)
    ingest_frame_views: bool = False
    ingest_result_ring: int = 16
    # process_evt3_words / process_evt2_words: events in front of a recording's first EVT_TIME_HIGH word are dropped (a reader that
    # waits for the first time base) instead of emitted at time base 0.  Which of the two Metavision's reader does is unpinned
    # (tools/pin_thirdparty.py decides); it matters for the first few words of a file only.
    raw_wait_for_time_base: bool = False

    @property
    def should_drop_frames(self):
        return not self.no_frame_dropping


def _enum_name(v) -> str:
    """'KEY_E' for UIKeyEvent.KEY_E, 'RELEASE' for UIAction.RELEASE, 'E' for "e", '69' for 69"""
    name = getattr(v, "name", None)
    if isinstance(name, str):
        return name.upper()
    return str(v).rsplit(".", 1)[-1].upper()


class FakeWindow:
    """Headless window: keeps the last frame so callers/tests can look at it."""

    def __init__(self):
        self.last_frame = None
        self.frames_shown = 0
        self._close = False

    def should_close(self):
        return self._close

    def set_close_flag(self):
        self._close = True

    def show_async(self, img):
        self.last_frame = img
        self.frames_shown += 1

    def set_keyboard_callback(self, cb):
        pass


@dataclass
class DepthReprojectionProcessor:
    params: RuntimeParams
    stats_printer: StatsPrinter = field(default_factory=StatsPrinter)
    window: Any = None  # anything with should_close() / show_async(img); default FakeWindow

    _pipe: DepthReprojectionPipe = field(init=False, default=None)
    _window: Any = field(init=False, default=None)

    def should_close(self):
        return self._window.should_close()

    def show_async(self, depth_map):
        self._window.show_async(depth_map)
        self.stats_printer.count("frames shown")

    def __enter__(self):
        self._pipe = DepthReprojectionPipe(params=self.params, stats_printer=self.stats_printer,
                                           frame_callback=self.show_async)
        self._window = self.window if self.window is not None else FakeWindow()
        if hasattr(self._window, "set_keyboard_callback"):
            self._window.set_keyboard_callback(self.keyboard_cb)
        return self

    def __exit__(self, *exc_info):
        self.stats_printer.print_stats()
        self._pipe.close()
        return False

    def keyboard_cb(self, key, scancode=None, action=None, mods=None):
        """Window key callback with Metavision's signature (key, scancode, action, mods), processor.py:96-105: acts on key
        RELEASE only; Q / Escape close, E selects the next frame event filter, S silences the statistics.  `key` / `action` may be
        metavision_sdk_ui's UIKeyEvent / UIAction enum members (compared by name, so the SDK need not be importable here), GLFW
        integers, or plain strings; a caller that passes no action (tests, a headless driver) means "released"."""
        if action is not None and _enum_name(action) not in ("RELEASE", "0"):  # (GLFW_RELEASE == 0)
            return
        k = _enum_name(key)
        if k in ("KEY_ESCAPE", "ESCAPE", "ESC", "256", "KEY_Q", "Q", "81"):
            self._window.set_close_flag()
        elif k in ("KEY_E", "E", "69"):
            self._pipe.select_next_frame_event_filter()
        elif k in ("KEY_S", "S", "83"):
            self.stats_printer.toggle_silence()

    def process_events(self, evs):
        self.stats_printer.print_stats_if_needed()
        self.stats_printer.count("processed evs", len(evs))
        self._pipe.process_events(evs)
        self.stats_printer.print_stats_if_needed()

    def process_evt3_words(self, words):
        """A chunk of a recording's EVT 3.0 words (x_maps_amd.evt3.read_raw_words) instead of an EventCD packet; with
        the device ingest (RuntimeParams.device_ingest, the default) the words are decoded on the device in front of it."""
        self.stats_printer.print_stats_if_needed()
        self._pipe.process_evt3_words(words)
        self.stats_printer.print_stats_if_needed()

    def process_evt2_words(self, words):
        """the same for a chunk of EVT 2.0 words (x_maps_amd.evt2.read_raw_words)"""
        self.stats_printer.print_stats_if_needed()
        self._pipe.process_evt2_words(words)
        self.stats_printer.print_stats_if_needed()

    def flush(self):
        """Device ingest: wait for the packets pushed so far and deliver the frames they produced."""
        self._pipe.flush()

    def reset(self):
        self._pipe.reset()
