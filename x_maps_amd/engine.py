"""XMapsEngine: thin, typed Python face of one xm_handle (include/xmaps.h).

All compute happens in libxmaps_hip.so on the GPU; this file only marshals NumPy arrays / raw device
pointers into the C-ABI and maps error codes to the exceptions the reference's NumPy code raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N

_T_DTYPES = {np.dtype(np.int64): N.XM_T_INT64, np.dtype(np.float32): N.XM_T_FLOAT32,
             np.dtype(np.float64): N.XM_T_FLOAT64}


@dataclass
class FrameStats:
    n_events: int = 0
    n_used: int = 0
    n_inliers: int = 0
    n_index_errors: int = 0
    t_min: float = 0.0
    t_max: float = 0.0
    gpu_ms: tuple = (0.0, 0.0, 0.0, 0.0)  # minmax, scatter, frame kernel, whole frame (profile only)
    n_unsorted: int = 0  # time-sorted mode: > 0 if the declaration did not hold for this frame

    @staticmethod
    def from_c(s: N.xm_frame_stats) -> "FrameStats":
        return FrameStats(int(s.n_events), int(s.n_used), int(s.n_inliers), int(s.n_index_errors),
                          float(s.t_min), float(s.t_max), tuple(float(v) for v in s.gpu_ms), int(s.n_unsorted))


def _ptr(a) -> C.c_void_p:
    if a is None:
        return C.c_void_p(None)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(int(a))  # raw device pointer


def _coords_u16(a, what: str) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.uint16:
        if a.size and (a.min() < 0 or a.max() > 65535):
            raise IndexError(f"{what} coordinate outside the sensor")
        a = a.astype(np.uint16)
    return np.ascontiguousarray(a)


def _time_col(t) -> tuple[np.ndarray, int]:
    t = np.asarray(t)
    if t.dtype not in _T_DTYPES:
        if np.issubdtype(t.dtype, np.integer):
            t = t.astype(np.int64)
        else:
            raise TypeError(f"unsupported time dtype {t.dtype}")
    return np.ascontiguousarray(t), _T_DTYPES[t.dtype]


def make_config(tables: dict, camera_perspective: bool = False, device: int = 0, n_slots: int = 1, flags: int = 0):
    """xm_config for `tables` (see XMapsEngine) + the arrays its pointers refer to (keep them alive until xm_create has returned)."""
    mapx = np.ascontiguousarray(tables["cam_mapx_i16"], dtype=np.int16)
    mapy = np.ascontiguousarray(tables["cam_mapy_i16"], dtype=np.int16)
    xmap = np.ascontiguousarray(tables["proj_x_map"], dtype=np.int16)
    pmap = tables.get("disp_proj_mapxy_i16")
    if pmap is not None:
        pmap = np.ascontiguousarray(pmap, dtype=np.int16)
    if mapx.shape != mapy.shape or mapx.ndim != 2 or xmap.ndim != 2:
        raise ValueError("bad table shapes")
    cam_h, cam_w = mapx.shape
    proj_h, proj_w = (pmap.shape[:2] if pmap is not None else (0, 0))
    cfg = N.xm_config()
    cfg.struct_size = C.sizeof(N.xm_config)
    cfg.device = device
    cfg.cam_width, cfg.cam_height = cam_w, cam_h
    cfg.proj_width, cfg.proj_height = proj_w, proj_h
    cfg.rect_width, cfg.rect_height = int(tables["rect_w"]), int(tables["rect_h"])
    cfg.xmap_height, cfg.xmap_width = xmap.shape
    cfg.x_offset = int(tables.get("x_offset", 4242))
    cfg.view = N.XM_VIEW_CAMERA if camera_perspective else N.XM_VIEW_PROJECTOR
    cfg.n_slots = n_slots
    cfg.flags = flags
    cfg.p03 = float(tables["p03"])
    cfg.z_near, cfg.z_far = float(tables["z_near"]), float(tables["z_far"])
    cfg.cam_mapx_i16 = mapx.ctypes.data
    cfg.cam_mapy_i16 = mapy.ctypes.data
    cfg.proj_x_map = xmap.ctypes.data
    cfg.disp_proj_mapxy_i16 = pmap.ctypes.data if pmap is not None else None
    return cfg, (mapx, mapy, xmap, pmap)


class XMapsEngine:
    """One GPU handle: tables resident in HBM + the fused per-frame kernels.

    tables: dict with the reference's arrays / scalars --
      cam_mapx_i16, cam_mapy_i16   (cam_h, cam_w) int16    CamProjMaps.disp_cam_map{x,y}_i16
      proj_x_map                   (H_x, W_t)     int16    XMapsDisparity.proj_x_map
      disp_proj_mapxy_i16          (proj_h, proj_w, 2) int16   CamProjMaps.disp_proj_mapxy_i16
      rect_w, rect_h, p03 (= P2[0,3]), z_near, z_far, x_offset (4242)
    """

    def __init__(self, tables: dict, camera_perspective: bool = False, device: int = 0, n_slots: int = 1,
                 assume_time_sorted: bool = False, try_sorted: bool = True, default_priority_streams: bool = False,
                 launch_workers: bool = False, force_general: bool = False, adaptive_batch: bool = False):
        """Extrema of t: by default the verified (t[0], t[n-1]) shortcut with automatic redo (exact for any order, the
        library's default); force_general=True (or try_sorted=False) runs the extrema pass K0 on every frame;
        assume_time_sorted=True declares the frames sorted (verified, reported instead of redone for asynchronous calls)."""
        self._lib = N.load_library()
        self._h = C.c_void_p(None)
        flags = ((N.XM_FLAG_TIME_SORTED if assume_time_sorted else 0)
                 | (N.XM_FLAG_GENERAL if (force_general or not try_sorted) else 0)
                 | (N.XM_FLAG_DEFAULT_STREAMS if default_priority_streams else 0)
                 | (N.XM_FLAG_LAUNCH_WORKERS if launch_workers else 0)
                 | (N.XM_FLAG_ADAPTIVE_BATCH if adaptive_batch else 0))
        cfg, keep = make_config(tables, camera_perspective, device, n_slots, flags)
        mapx, xmap, pmap = keep[0], keep[2], keep[3]
        cam_h, cam_w = mapx.shape
        proj_h, proj_w = (pmap.shape[:2] if pmap is not None else (0, 0))
        self.p03 = cfg.p03
        N.check(self._lib.xm_create(C.byref(cfg), C.byref(self._h)))
        self._pinned = []
        self.camera_perspective = camera_perspective
        self.device = device
        self.n_slots = n_slots
        self.cam_w, self.cam_h, self.proj_w, self.proj_h = cam_w, cam_h, proj_w, proj_h
        self.rect_w, self.rect_h = cfg.rect_width, cfg.rect_height
        self.out_h, self.out_w = (cam_h, cam_w) if camera_perspective else (proj_h, proj_w)
        # packed-key frame as laid out in HBM: camera view row-major [y][x]; projector view column-major [col][row]
        self.key_shape = (cam_h, cam_w) if camera_perspective else (self.rect_w, self.rect_h)
        self.t_px_scale = xmap.shape[1] - 1

    # ---- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            for p in getattr(self, "_pinned", []):
                self._lib.xm_host_free(self._h, C.c_void_p(p))
            self._pinned = []
            self._lib.xm_destroy(self._h)
            self._h = C.c_void_p(None)

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

    def sync(self):
        N.check(self._lib.xm_sync(self._h))

    def sorted_fallbacks(self) -> int:
        """Frames redone on the general path because a time-sorted shortcut did not hold."""
        v = C.c_uint64(0)
        N.check(self._lib.xm_sorted_fallbacks(self._h, C.byref(v)))
        return int(v.value)

    def path_counts(self) -> dict:
        """Frames enqueued per K1 variant since the engine was created (a redone frame counts twice)."""
        v = (C.c_uint64 * 4)()
        N.check(self._lib.xm_path_counts(self._h, v))
        return {"general": int(v[0]), "sorted_key64": int(v[1]), "key32": int(v[2]), "cols": int(v[3])}

    def cols_info(self) -> dict:
        """Which no-atomics K1 the rig qualified for (xm_cols_info): mode 'none' / 'cols' (injective X-map) / 'own' (several
        time columns per frame cell: the reference's own calibration) and the owner tiles' geometry."""
        a = (C.c_int32 * 12)()
        N.check(self._lib.xm_cols_info(self._h, a))
        return {"mode": ("none", "cols", "own")[a[0]], "w": a[1], "halo": a[2], "nxs_max": a[3], "shear_m": a[4],
                "shear_extra": a[5], "r_lo": a[6], "rows": a[7], "extras": a[8], "extras_max_per_tile": a[9],
                "dense_w": a[10], "dense_halo": a[11]}  # (the second plan: narrow tiles for frames too dense for the first; 0 = none)

    def stream(self, slot: int = 0) -> int:
        return int(self._lib.xm_stream(self._h, slot) or 0)

    # ---- fused hot path, host arrays ------------------------------------------------------------
    def process_frame(self, x, y, t, p=None, want_depth=True, want_bgr=True, raise_on_index_error=True):
        """SoA events (host) -> (depth f32 [H,W] | None, bgr u8 [H,W,3] | None, FrameStats)."""
        x, y = _coords_u16(x, "x"), _coords_u16(y, "y")
        t, tdt = _time_col(t)
        n = len(t)
        if len(x) != n or len(y) != n:
            raise ValueError("x, y, t must have equal length")
        if p is not None:
            p = np.ascontiguousarray(p, dtype=np.int16)
        depth = np.empty((self.out_h, self.out_w), np.float32) if want_depth else None
        bgr = np.empty((self.out_h, self.out_w, 3), np.uint8) if want_bgr else None
        st = N.xm_frame_stats()
        rc = self._lib.xm_process_frame(self._h, _ptr(x), _ptr(y), _ptr(t), _ptr(p), n, tdt, N.XM_MEM_HOST,
                                        _ptr(depth), _ptr(bgr), C.byref(st))
        N.check(rc, index_error_ok=not raise_on_index_error)
        return depth, bgr, FrameStats.from_c(st)

    def process_events(self, evs: np.ndarray, use_polarity=False, want_depth=True, want_bgr=True,
                       raise_on_index_error=True):
        """Metavision EventCD structured array (16-byte AoS records) -> (depth, bgr, FrameStats)."""
        f = evs.dtype.fields
        if f is None or not {"x", "y", "p", "t"} <= set(f):
            raise TypeError("expected a structured EventCD array with fields x, y, p, t")
        if evs.dtype.itemsize != 16 or (f["x"][1], f["y"][1], f["p"][1], f["t"][1]) != (0, 2, 4, 8) or \
                (f["x"][0], f["y"][0], f["p"][0], f["t"][0]) != (np.uint16, np.uint16, np.int16, np.int64):
            # e.g. np.concatenate of EventCD buffers under NumPy 2 returns the PACKED 14-byte layout
            from .synthetic import EVENT_CD_DTYPE
            evs = evs.astype(EVENT_CD_DTYPE)
        evs = np.ascontiguousarray(evs)
        depth = np.empty((self.out_h, self.out_w), np.float32) if want_depth else None
        bgr = np.empty((self.out_h, self.out_w, 3), np.uint8) if want_bgr else None
        st = N.xm_frame_stats()
        rc = self._lib.xm_process_frame_aos(self._h, _ptr(evs) if len(evs) else None, len(evs), int(use_polarity),
                                            N.XM_MEM_HOST, _ptr(depth), _ptr(bgr), C.byref(st))
        N.check(rc, index_error_ok=not raise_on_index_error)
        return depth, bgr, FrameStats.from_c(st)

    # ---- fused hot path, device pointers (async) -------------------------------------------------
    def process_frame_device(self, x_ptr, y_ptr, t_ptr, p_ptr, n, depth_ptr=None, bgr_ptr=None,
                             t_dtype=N.XM_T_INT64):
        """Asynchronous: returns once the frame's kernels are enqueued on one of the handle's n_slots streams.  CONTRACT of the
        default (verified-shortcut) mode: the frame's inputs AND outputs must stay untouched until `sync()` has returned or
        `n_slots` further frames have been submitted -- a frame whose (t[0], t[n-1]) / tile verification failed is redone from its
        inputs into its outputs at that point, so a caller that only synchronises `stream()` can read a not-yet-redone frame.
        (force_general=True or assume_time_sorted=True never redo: there the stream alone orders the outputs.)"""
        N.check(self._lib.xm_process_frame(self._h, _ptr(x_ptr), _ptr(y_ptr), _ptr(t_ptr), _ptr(p_ptr), n, t_dtype,
                                           N.XM_MEM_DEVICE, _ptr(depth_ptr), _ptr(bgr_ptr), None))

    def process_events_device(self, aos_ptr, n, use_polarity=False, depth_ptr=None, bgr_ptr=None):
        """EventCD records (16-byte AoS) resident on the device; asynchronous, same contract as process_frame_device."""
        N.check(self._lib.xm_process_frame_aos(self._h, _ptr(aos_ptr), n, int(use_polarity), N.XM_MEM_DEVICE,
                                               _ptr(depth_ptr), _ptr(bgr_ptr), None))

    def profile_frame_device(self, x_ptr, y_ptr, t_ptr, p_ptr, n, depth_ptr=None, bgr_ptr=None,
                             t_dtype=N.XM_T_INT64) -> FrameStats:
        st = N.xm_frame_stats()
        N.check(self._lib.xm_profile_frame(self._h, _ptr(x_ptr), _ptr(y_ptr), _ptr(t_ptr), _ptr(p_ptr), n, t_dtype,
                                           _ptr(depth_ptr), _ptr(bgr_ptr), C.byref(st)), index_error_ok=True)
        return FrameStats.from_c(st)

    def profile_event_overhead_ms(self, reps: int = 25) -> float:
        ms = C.c_float(0)
        N.check(self._lib.xm_profile_event_overhead(self._h, reps, C.byref(ms)))
        return float(ms.value)

    def last_frame_stats(self) -> FrameStats:
        st = N.xm_frame_stats()
        N.check(self._lib.xm_last_frame_stats(self._h, C.byref(st)))
        return FrameStats.from_c(st)

    # ---- a group of frames in one set of multi-frame launches ------------------------------------------
    def process_batch_device(self, x_ptr, y_ptr, t_ptr, p_ptr, offsets, depth_ptr=None, bgr_ptr=None,
                             t_dtype=N.XM_T_INT64):
        """Frames [offsets[f], offsets[f+1]) of device-resident SoA columns -> depth_ptr + f*H*W (bgr_ptr + f*H*W*3);
        len(offsets) - 1 <= n_slots; asynchronous (sync() to wait)."""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        N.check(self._lib.xm_process_batch(self._h, _ptr(x_ptr), _ptr(y_ptr), _ptr(t_ptr), _ptr(p_ptr), t_dtype,
                                           offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(offs) - 1,
                                           _ptr(depth_ptr), _ptr(bgr_ptr)))

    def process_events_batch_device(self, aos_ptr, offsets, depth_ptr=None, bgr_ptr=None):
        """Frames [offsets[f], offsets[f+1]) of device-resident EventCD records (16-byte AoS); asynchronous."""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        N.check(self._lib.xm_process_batch_aos(self._h, _ptr(aos_ptr), offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(offs) - 1,
                                               _ptr(depth_ptr), _ptr(bgr_ptr)))

    def process_event_frames(self, frames, want_depth=True, want_bgr=True):
        """A list of EventCD frames (host arrays, as the trigger finder cuts them) in groups of <= n_slots frames through the
        multi-frame launches: [(depth | None, bgr | None)] per frame.  Synchronous; what an offline replay uses instead of one
        call per frame (the kernels of a group keep the chip full: DESIGN.md section 3)."""
        from .synthetic import EVENT_CD_DTYPE
        out = []
        px = self.out_h * self.out_w
        G = max(1, self.n_slots)
        for g0 in range(0, len(frames), G):
            grp = [np.ascontiguousarray(f if f.dtype == EVENT_CD_DTYPE else f.astype(EVENT_CD_DTYPE)) for f in frames[g0:g0 + G]]
            if any(len(f) == 0 for f in grp):
                raise ValueError("empty frame in a group (t.min() of an empty frame raises in the reference)")
            offs = np.zeros(len(grp) + 1, np.uint64)
            offs[1:] = np.cumsum([len(f) for f in grp])
            rec = np.empty(int(offs[-1]), EVENT_CD_DTYPE)  # (np.concatenate would hand back the packed 14-byte layout under NumPy 2)
            for i, f in enumerate(grp):
                rec[int(offs[i]):int(offs[i + 1])] = f
            d_rec = self.to_device(rec)
            d_depth = self.dev_alloc(len(grp) * px * 4) if want_depth else None
            d_bgr = self.dev_alloc(len(grp) * px * 3) if want_bgr else None
            try:
                self.process_events_batch_device(d_rec, offs, d_depth, d_bgr)
                self.sync()  # (also redoes frames whose verified shortcut failed)
                depth = np.empty((len(grp), self.out_h, self.out_w), np.float32) if want_depth else None
                bgr = np.empty((len(grp), self.out_h, self.out_w, 3), np.uint8) if want_bgr else None
                if want_depth:
                    self.dev_download(depth, d_depth)
                if want_bgr:
                    self.dev_download(bgr, d_bgr)
            finally:
                for ptr in (d_rec, d_depth, d_bgr):
                    if ptr:
                        self.dev_free(ptr)
            out += [(None if depth is None else depth[i], None if bgr is None else bgr[i]) for i in range(len(grp))]
        return out

    def profile_batch_device(self, x_ptr, y_ptr, t_ptr, p_ptr, offsets, depth_ptr=None, bgr_ptr=None,
                             t_dtype=N.XM_T_INT64):
        """The same group, synchronously, timed per launch: (K0 or K0b, K1, K2, first start .. last stop) in ms."""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        ms = (C.c_float * 4)()
        N.check(self._lib.xm_profile_batch(self._h, _ptr(x_ptr), _ptr(y_ptr), _ptr(t_ptr), _ptr(p_ptr), t_dtype,
                                           offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(offs) - 1,
                                           _ptr(depth_ptr), _ptr(bgr_ptr), ms))
        return tuple(float(v) for v in ms)

    # ---- hipGraph batch ----------------------------------------------------------------------------
    def graph_create(self, x_ptr, y_ptr, t_ptr, p_ptr, offsets, depth_ptr=None, bgr_ptr=None,
                     t_dtype=N.XM_T_INT64) -> "XMapsGraph":
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        g = C.c_void_p(None)
        N.check(self._lib.xm_graph_create(self._h, _ptr(x_ptr), _ptr(y_ptr), _ptr(t_ptr), _ptr(p_ptr), t_dtype,
                                          offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(offs) - 1,
                                          _ptr(depth_ptr), _ptr(bgr_ptr), C.byref(g)))
        return XMapsGraph(self, g, len(offs) - 1)

    def debug_cols_thresholds(self, t_first: int, t_last: int) -> np.ndarray:
        """thr[0 .. xmap_w] of the column-tile path for a frame with these first / last stamps (tests)."""
        out = np.zeros(self.t_px_scale + 2, np.uint32)
        N.check(self._lib.xm_debug_cols_thresholds(self._h, int(t_first), int(t_last), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    # ---- debug: every per-event intermediate -------------------------------------------------------
    def debug_k2_pipe_frames(self) -> int:
        """tests: frames finished by the software-pipelined K2 since the engine was created"""
        v = C.c_uint64(0)
        N.check(self._lib.xm_debug_k2_pipe_frames(self._h, C.byref(v)))
        return int(v.value)

    def debug_last_disp_frame(self) -> np.ndarray:
        """tests: A3's u16 disparity frame [rect_h][rect_w] of the last frame, when it took the column / owner tiles"""
        out = np.zeros((self.rect_h, self.rect_w), np.uint16)
        N.check(self._lib.xm_debug_last_disp_frame(self._h, out.ctypes.data_as(C.POINTER(C.c_uint16))))
        return out

    def debug_event_outputs(self, x, y, t, p=None):
        x, y = _coords_u16(x, "x"), _coords_u16(y, "y")
        t, tdt = _time_col(t)
        n = len(t)
        if p is not None:
            p = np.ascontiguousarray(p, dtype=np.int16)
        out = {k: np.zeros(n, np.int16) for k in ("xr", "yr", "ts", "disp")}
        out["mask"] = np.zeros(n, np.uint8)
        N.check(self._lib.xm_debug_event_outputs(self._h, _ptr(x), _ptr(y), _ptr(t), _ptr(p), n, tdt, N.XM_MEM_HOST,
                                                 _ptr(out["xr"]), _ptr(out["yr"]), _ptr(out["ts"]),
                                                 _ptr(out["disp"]), _ptr(out["mask"])))
        out["mask"] = out["mask"].astype(bool)
        return out

    # ---- the reference's stages one by one ----------------------------------------------------------
    def rectify_cam_coords_i16(self, x, y):
        x, y = _coords_u16(x, "x"), _coords_u16(y, "y")
        xr = np.empty(len(x), np.int16)
        yr = np.empty(len(x), np.int16)
        N.check(self._lib.xm_stage_rectify(self._h, _ptr(x), _ptr(y), len(x), _ptr(xr), _ptr(yr)))
        return xr, yr

    def rectify_cam_coords_f32(self, mapx_f32, mapy_f32, x, y):
        """CamProjMaps.rectify_cam_coords_f32 (cam_proj_calibration.py:272-275) from float maps [cam_h][cam_w]."""
        mx = np.ascontiguousarray(mapx_f32, dtype=np.float32)
        my = np.ascontiguousarray(mapy_f32, dtype=np.float32)
        if mx.shape != (self.cam_h, self.cam_w) or my.shape != mx.shape:
            raise ValueError(f"float rectify maps must be ({self.cam_h}, {self.cam_w})")
        x, y = _coords_u16(x, "x"), _coords_u16(y, "y")
        xr = np.empty(len(x), np.float32)
        yr = np.empty(len(x), np.float32)
        N.check(self._lib.xm_stage_rectify_f32(self._h, _ptr(mx), _ptr(my), _ptr(x), _ptr(y), len(x), _ptr(xr), _ptr(yr)))
        return xr, yr

    def construct_point_cloud(self, Q, xpr_f32, ypr_f32, disp_f32):
        """CamProjMaps.construct_point_cloud (cam_proj_calibration.py:319-331) -> float32 [n][3]."""
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        if Q.shape != (4, 4):
            raise ValueError("Q must be 4x4")
        xp = np.ascontiguousarray(xpr_f32, dtype=np.float32)
        yp = np.ascontiguousarray(ypr_f32, dtype=np.float32)
        d = np.ascontiguousarray(disp_f32, dtype=np.float32)
        if not (len(xp) == len(yp) == len(d)):
            raise ValueError("xpr, ypr and disp must have the same length")
        cloud = np.empty((len(d), 3), np.float32)
        N.check(self._lib.xm_stage_point_cloud(self._h, _ptr(Q), _ptr(xp), _ptr(yp), _ptr(d), len(d), _ptr(cloud)))
        return cloud

    def event_disparity_full(self, xr_i16, yr_i16, t):
        """A2 with full-length outputs: (disp[n] int16 with 0 where masked, mask[n] bool)."""
        xr = np.ascontiguousarray(xr_i16, dtype=np.int16)
        yr = np.ascontiguousarray(yr_i16, dtype=np.int16)
        t, tdt = _time_col(t)
        n = len(t)
        if n == 0:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        disp = np.empty(n, np.int16)
        mask = np.empty(n, np.uint8)
        N.check(self._lib.xm_stage_event_disparity(self._h, _ptr(xr), _ptr(yr), _ptr(t), n, tdt, _ptr(disp), _ptr(mask)))
        return disp, mask.astype(bool)

    def disp_map_projector_view(self, xr_i16, yr_i16, disp_full, mask):
        xr = np.ascontiguousarray(xr_i16, dtype=np.int16)
        yr = np.ascontiguousarray(yr_i16, dtype=np.int16)
        d = np.ascontiguousarray(disp_full, dtype=np.int16)
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        out = np.empty((self.rect_h, self.rect_w), np.float32)
        N.check(self._lib.xm_stage_disp_map_projector_view(self._h, _ptr(xr), _ptr(yr), _ptr(d), _ptr(m), len(d), _ptr(out)))
        return out

    def disp_map_camera_view(self, x, y, disp_full, mask):
        x, y = _coords_u16(x, "x"), _coords_u16(y, "y")
        d = np.ascontiguousarray(disp_full, dtype=np.int16)
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        out = np.empty((self.cam_h, self.cam_w), np.float32)
        N.check(self._lib.xm_stage_disp_map_camera_view(self._h, _ptr(x), _ptr(y), _ptr(d), _ptr(m), len(d), _ptr(out)))
        return out

    def remap_rectified_disp_map_to_proj(self, rect_disp):
        src = np.ascontiguousarray(rect_disp, dtype=np.float32)
        if src.shape != (self.rect_h, self.rect_w):
            raise ValueError(f"expected a {(self.rect_h, self.rect_w)} frame, got {src.shape}")
        out = np.empty((self.proj_h, self.proj_w), np.float32)
        N.check(self._lib.xm_stage_remap_rectified_disp_map_to_proj(self._h, _ptr(src), _ptr(out)))
        return out

    def disparity_to_depth(self, disp):
        src = np.ascontiguousarray(disp, dtype=np.float32)
        out = np.empty(src.shape, np.float32)
        N.check(self._lib.xm_stage_disparity_to_depth(self._h, _ptr(src), src.shape[0], src.shape[1], _ptr(out)))
        return out

    def colorize_depth_from_disp(self, disp):
        src = np.ascontiguousarray(disp, dtype=np.float32)
        out = np.empty(src.shape + (3,), np.uint8)
        N.check(self._lib.xm_stage_colorize_depth_from_disp(self._h, _ptr(src), src.shape[0], src.shape[1], _ptr(out)))
        return out

    # ---- N3: per-frame de-duplication filters ------------------------------------------------------------
    def frame_event_filter(self, filter_id: int, evs: np.ndarray, xp_i16=None, map_shape=None,
                           intended_semantics: bool = False) -> np.ndarray:
        from .synthetic import EVENT_CD_DTYPE
        if evs.dtype != EVENT_CD_DTYPE:
            evs = evs.astype(EVENT_CD_DTYPE)
        evs = np.ascontiguousarray(evs)
        h, w = map_shape
        out = np.zeros(max(int(h) * int(w), 1), dtype=EVENT_CD_DTYPE)
        n_out = C.c_size_t(0)
        xp = None if xp_i16 is None else np.ascontiguousarray(xp_i16, dtype=np.int16)
        N.check(self._lib.xm_frame_event_filter(self._h, int(filter_id), int(intended_semantics),
                                                _ptr(evs) if len(evs) else None, len(evs), _ptr(xp),
                                                int(h), int(w), _ptr(out), C.byref(n_out)))
        return out[:n_out.value].copy()

    # ---- N2: pause detection on the device -------------------------------------------------------------------
    def find_pauses(self, t=None, evs=None, thresh_us: int = 40, device_ptr=None, n=None, aos=False) -> np.ndarray:
        """np.nonzero(np.diff(t) >= thresh_us)[0] (python/trigger_finder.py:155).  Host arrays (t int64 or EventCD evs) or a
        device pointer (device_ptr, n, aos=True for EventCD records)."""
        if device_ptr is not None:
            cap = int(n)
            mem, tp, ap = N.XM_MEM_DEVICE, (None if aos else device_ptr), (device_ptr if aos else None)
        elif evs is not None:
            from .synthetic import EVENT_CD_DTYPE
            evs = np.ascontiguousarray(evs if evs.dtype == EVENT_CD_DTYPE else evs.astype(EVENT_CD_DTYPE))
            cap, mem, tp, ap = len(evs), N.XM_MEM_HOST, None, evs
        else:
            t = np.ascontiguousarray(t, dtype=np.int64)
            cap, mem, tp, ap = len(t), N.XM_MEM_HOST, t, None
        out = np.empty(max(cap, 1), np.uint32)
        n_out = C.c_size_t(0)
        N.check(self._lib.xm_find_pauses(self._h, _ptr(tp), _ptr(ap), cap, mem, int(thresh_us), _ptr(out), len(out), C.byref(n_out)))
        return out[:n_out.value].astype(np.int64)

    # ---- shards (device pointers) ----------------------------------------------------------------------
    def shard_minmax(self, t_ptr, p_ptr, n, t_dtype=N.XM_T_INT64):
        np_dt = {N.XM_T_INT64: np.int64, N.XM_T_FLOAT32: np.float32, N.XM_T_FLOAT64: np.float64}[t_dtype]
        out = np.zeros(2, np_dt)
        N.check(self._lib.xm_shard_minmax(self._h, _ptr(t_ptr), _ptr(p_ptr), n, t_dtype, _ptr(out)))
        return out

    def shard_minmax_device(self, t_ptr, p_ptr, n, mm_dev_ptr, t_dtype=N.XM_T_INT64):
        """Shard extrema -> {tmin, -tmax} (int64 / float64) in a 16-byte device buffer; no host synchronisation."""
        N.check(self._lib.xm_shard_minmax_device(self._h, _ptr(t_ptr), _ptr(p_ptr), n, t_dtype, _ptr(mm_dev_ptr)))

    def shard_scatter_device(self, x_ptr, y_ptr, t_ptr, p_ptr, n, idx_offset, mm_dev_ptr, tag, key_ptr,
                             t_dtype=N.XM_T_INT64):
        N.check(self._lib.xm_shard_scatter_device(self._h, _ptr(x_ptr), _ptr(y_ptr), _ptr(t_ptr), _ptr(p_ptr), n, t_dtype,
                                                  int(idx_offset), _ptr(mm_dev_ptr), int(tag), _ptr(key_ptr)))

    def shard_clear(self, key_ptr):
        N.check(self._lib.xm_shard_clear(self._h, _ptr(key_ptr)))

    def shard_scatter(self, x_ptr, y_ptr, t_ptr, p_ptr, n, idx_offset, frame_minmax, tag, key_ptr,
                      t_dtype=N.XM_T_INT64):
        np_dt = {N.XM_T_INT64: np.int64, N.XM_T_FLOAT32: np.float32, N.XM_T_FLOAT64: np.float64}[t_dtype]
        mm = np.ascontiguousarray(frame_minmax, dtype=np_dt)
        N.check(self._lib.xm_shard_scatter(self._h, _ptr(x_ptr), _ptr(y_ptr), _ptr(t_ptr), _ptr(p_ptr), n, t_dtype,
                                           int(idx_offset), _ptr(mm), int(tag), _ptr(key_ptr)))

    def shard_finish(self, key_ptr, tag, depth_ptr=None, bgr_ptr=None):
        N.check(self._lib.xm_shard_finish(self._h, _ptr(key_ptr), int(tag), _ptr(depth_ptr), _ptr(bgr_ptr)))

    def shard_decode_u16(self, key_ptr, n_cells, tag, out_ptr):
        N.check(self._lib.xm_shard_decode_u16(self._h, _ptr(key_ptr), int(n_cells), int(tag), _ptr(out_ptr)))

    def shard_finish_u16_band(self, disp_ptr, col_lo: int, col_hi: int, depth_ptr=None, bgr_ptr=None):
        """The frame kernel for the projector tiles centred on frame columns [col_lo, col_hi) only (band-sharded finish)."""
        N.check(self._lib.xm_shard_finish_u16_band(self._h, _ptr(disp_ptr), int(col_lo), int(col_hi), _ptr(depth_ptr), _ptr(bgr_ptr)))

    def k2_patch_cols_max(self) -> int:
        v = C.c_int(0)
        N.check(self._lib.xm_k2_patch_cols_max(self._h, C.byref(v)))
        return int(v.value)

    def shard_finish_u16(self, disp_ptr, depth_ptr=None, bgr_ptr=None):
        N.check(self._lib.xm_shard_finish_u16(self._h, _ptr(disp_ptr), _ptr(depth_ptr), _ptr(bgr_ptr)))

    # ---- shards on the column tiles (include/xmaps.h: xm_shard_cols_*) ----
    def shard_cols_last_k1_ms(self) -> float:
        """milliseconds of the column-tile K1 alone in the last shard_cols_scatter issued under XM_SHARD_PROFILE (synchronises)"""
        ms = C.c_float(0.0)
        N.check(self._lib.xm_shard_cols_last_k1_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def shard_cols_info(self, n_frame_events):
        """{frame_bytes, reduce_u32, send_bytes, cap_events} for frames of that many events, or None when the rig / the density
        does not take the column tiles"""
        fb, ru, sb, ce = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = self._lib.xm_shard_cols_info(self._h, int(n_frame_events), C.byref(fb), C.byref(ru), C.byref(sb), C.byref(ce))
        if rc != 0:
            return None
        return {"frame_bytes": int(fb.value), "reduce_u32": int(ru.value), "send_bytes": int(sb.value), "cap_events": int(ce.value)}

    def shard_cols_pack(self, x_ptr, y_ptr, t_ptr, n, send_ptr, cap_events):
        N.check(self._lib.xm_shard_cols_pack(self._h, _ptr(x_ptr), _ptr(y_ptr), _ptr(t_ptr), int(n), _ptr(send_ptr), int(cap_events)))

    def shard_cols_scatter(self, x_ptr, y_ptr, t_ptr, n, n_frame_events, gathered_ptr, send_bytes, rank, world, cap_events, frame16_ptr):
        N.check(self._lib.xm_shard_cols_scatter(self._h, _ptr(x_ptr), _ptr(y_ptr), _ptr(t_ptr), int(n), int(n_frame_events),
                                                _ptr(gathered_ptr), int(send_bytes), int(rank), int(world), int(cap_events), _ptr(frame16_ptr)))

    def shard_cols_failed(self) -> bool:
        v = C.c_int(0)
        N.check(self._lib.xm_shard_cols_failed(self._h, C.byref(v)))
        return bool(v.value)

    # ---- pinned host memory + asynchronous host path ---------------------------------------------------------
    def host_empty(self, shape, dtype) -> np.ndarray:
        """NumPy array backed by pinned host memory (freed with the engine).  For process_frame_pinned."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p(None)
        N.check(self._lib.xm_host_alloc(self._h, max(n, 16), C.byref(p)))
        self._pinned.append(p.value)
        buf = (C.c_char * max(n, 16)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def process_frame_pinned(self, x, y, t, p=None, depth_out=None, bgr_out=None):
        """Asynchronous frame from PINNED host arrays (host_empty): H2D, kernels and D2H enqueued on the next slot's
        stream; returns at once; sync() before reading depth_out / bgr_out or reusing the inputs."""
        _, tdt = _time_col(t)
        N.check(self._lib.xm_process_frame(self._h, _ptr(x), _ptr(y), _ptr(t), _ptr(p), len(t), tdt, N.XM_MEM_HOST_PINNED,
                                           _ptr(depth_out), _ptr(bgr_out), None))

    def process_events_pinned(self, evs, depth_out=None, bgr_out=None, use_polarity=False):
        N.check(self._lib.xm_process_frame_aos(self._h, _ptr(evs), len(evs), int(use_polarity), N.XM_MEM_HOST_PINNED,
                                               _ptr(depth_out), _ptr(bgr_out), None))

    # ---- device memory helpers (for hosts without torch) -------------------------------------------------
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p(None)
        N.check(self._lib.xm_dev_alloc(self._h, nbytes, C.byref(p)))
        return int(p.value)

    def dev_free(self, ptr: int):
        N.check(self._lib.xm_dev_free(self._h, _ptr(ptr)))

    def dev_upload(self, ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        N.check(self._lib.xm_dev_upload(self._h, _ptr(ptr), _ptr(arr), arr.nbytes))

    def dev_download(self, arr: np.ndarray, ptr: int):
        assert arr.flags.c_contiguous
        N.check(self._lib.xm_dev_download(self._h, _ptr(arr), _ptr(ptr), arr.nbytes))

    def to_device(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = self.dev_alloc(arr.nbytes)
        self.dev_upload(p, arr)
        return p


class XMapsGraph:
    def __init__(self, engine: XMapsEngine, handle: C.c_void_p, n_frames: int):
        self._e, self._g, self.n_frames = engine, handle, n_frames

    def launch(self):
        N.check(self._e._lib.xm_graph_launch(self._g))

    def close(self):
        if self._g is not None and self._g.value:
            self._e._lib.xm_graph_destroy(self._g)
            self._g = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
