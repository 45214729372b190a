"""ctypes binding of libxmaps_hip.so (the C-ABI in include/xmaps.h) + the in-tree build recipe.

There is NO CPU fallback: if the library is missing and cannot be built, or no AMD GPU is visible,
the product path raises.  (The CPU restatement under oracle/ is test infrastructure only.)
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "libxmaps_hip.so")
SOURCES = [os.path.join(PKG_DIR, "csrc", "xmaps_hip.hip")]
def _depends():
    """every file the one translation unit is made of: csrc/*.hpp, csrc/*.inc, csrc/host/*.hpp, include/xmaps.h"""
    import glob
    csrc = os.path.join(PKG_DIR, "csrc")
    return SOURCES + sorted(glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(csrc, "*.inc")) +
                            glob.glob(os.path.join(csrc, "host", "*.hpp"))) + [os.path.join(ROOT, "include", "xmaps.h")]


DEPENDS = _depends()
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]

XM_OK, XM_ERR_INVALID, XM_ERR_HIP, XM_ERR_NOMEM, XM_ERR_INDEX, XM_ERR_TOO_MANY, XM_ERR_UNSORTED = 0, -1, -2, -3, -4, -5, -6
XM_FLAG_TIME_SORTED = 1
XM_FLAG_TRY_SORTED = 2
XM_FLAG_DEFAULT_STREAMS = 4
XM_FLAG_LAUNCH_WORKERS = 8
XM_FLAG_GENERAL = 16
XM_FLAG_ADAPTIVE_BATCH = 32
XM_VIEW_PROJECTOR, XM_VIEW_CAMERA = 0, 1
XM_MEM_HOST, XM_MEM_DEVICE, XM_MEM_HOST_PINNED = 0, 1, 2
XM_T_INT64, XM_T_FLOAT32, XM_T_FLOAT64 = 0, 1, 2


class XMapsNativeError(RuntimeError):
    pass


class XMapsTooMany(ValueError):
    """XM_ERR_TOO_MANY: more events / words than the buffer the call was given can hold (nothing has advanced: hand the input
    over again in smaller pieces)"""


def _hipcc() -> str | None:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    so_m = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(d) and os.path.getmtime(d) > so_m for d in DEPENDS)


def build_native(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 ... -> x_maps_amd/libxmaps_hip.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    if hipcc is None:
        raise XMapsNativeError("hipcc not found: cannot build libxmaps_hip.so (and there is no CPU fallback)")
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = [hipcc] + HIPCC_FLAGS + SOURCES + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise XMapsNativeError("hipcc failed:\n" + r.stderr[-4000:])
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


class xm_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32),
        ("cam_width", C.c_int32), ("cam_height", C.c_int32),
        ("proj_width", C.c_int32), ("proj_height", C.c_int32),
        ("rect_width", C.c_int32), ("rect_height", C.c_int32),
        ("xmap_width", C.c_int32), ("xmap_height", C.c_int32),
        ("x_offset", C.c_int32), ("view", C.c_int32), ("n_slots", C.c_int32), ("flags", C.c_uint32),
        ("p03", C.c_double), ("z_near", C.c_float), ("z_far", C.c_float),
        ("cam_mapx_i16", C.c_void_p), ("cam_mapy_i16", C.c_void_p),
        ("proj_x_map", C.c_void_p), ("disp_proj_mapxy_i16", C.c_void_p),
    ]


class xm_frame_stats(C.Structure):
    _fields_ = [
        ("n_events", C.c_uint64), ("n_used", C.c_uint64), ("n_inliers", C.c_uint64),
        ("n_index_errors", C.c_uint64), ("t_min", C.c_double), ("t_max", C.c_double),
        ("gpu_ms", C.c_float * 4), ("n_unsorted", C.c_uint64),
    ]


XM_INGEST_NO_LAUNCH_THREAD = 1
XM_INGEST_ACT_SELF = 2


class xm_ingest_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("projector_fps", C.c_int32), ("use_polarity", C.c_int32), ("activity_filter", C.c_int32),
        ("activity_thresh_us", C.c_int64), ("pause_thresh_us", C.c_int64),
        ("min_events_per_frame", C.c_int32), ("result_ring", C.c_int32),
        ("capacity_events", C.c_uint64), ("max_packet_events", C.c_uint64), ("expected_events_per_frame", C.c_uint64),
        ("want_depth", C.c_int32), ("want_bgr", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_uint32),
    ]


class xm_ingest_frame(C.Structure):
    _fields_ = [
        ("seq", C.c_uint64), ("n_events", C.c_uint64), ("t_first", C.c_int64), ("t_last", C.c_int64),
        ("n_inliers", C.c_uint64), ("n_index_errors", C.c_uint64), ("live_after", C.c_uint64),
        ("overflow", C.c_uint32), ("lost", C.c_uint32), ("depth", C.c_void_p), ("bgr", C.c_void_p), ("push_seq", C.c_uint64),
        ("push_to_publish_us", C.c_float), ("owned", C.c_uint32),
    ]


class xm_eval_result(C.Structure):
    _fields_ = [("fillrate", C.c_double), ("rmse", C.c_double), ("perc_1", C.c_double), ("perc_5", C.c_double),
                ("perc_10", C.c_double), ("margin", C.c_double), ("n_valid", C.c_uint64), ("n_gt_zero", C.c_uint64)]


# every symbol include/xmaps.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "xm_api_version": (C.c_int, []),
    "xm_debug_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "xm_last_error": (C.c_char_p, []),
    "xm_create": (C.c_int, [C.POINTER(xm_config), C.POINTER(_P)]),
    "xm_destroy": (None, [_P]),
    "xm_sync": (C.c_int, [_P]),
    "xm_sorted_fallbacks": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "xm_path_counts": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "xm_cols_info": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "xm_own_plan_info": (C.c_int, [C.POINTER(xm_config), C.POINTER(C.c_int32)]),
    "xm_process_frame": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int, C.c_int, _P, _P, C.POINTER(xm_frame_stats)]),
    "xm_process_frame_aos": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P, _P, C.POINTER(xm_frame_stats)]),
    "xm_last_frame_stats": (C.c_int, [_P, C.POINTER(xm_frame_stats)]),
    "xm_profile_frame": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int, _P, _P, C.POINTER(xm_frame_stats)]),
    "xm_profile_event_overhead": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    "xm_process_batch": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_uint64), C.c_int, _P, _P]),
    "xm_process_batch_aos": (C.c_int, [_P, _P, C.POINTER(C.c_uint64), C.c_int, _P, _P]),
    "xm_profile_batch": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_uint64), C.c_int, _P, _P, C.POINTER(C.c_float)]),
    "xm_graph_create": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_uint64), C.c_int, _P, _P, C.POINTER(_P)]),
    "xm_graph_launch": (C.c_int, [_P]),
    "xm_graph_destroy": (None, [_P]),
    "xm_debug_k2_pipe_frames": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "xm_debug_last_disp_frame": (C.c_int, [_P, C.POINTER(C.c_uint16)]),
    "xm_debug_cols_thresholds": (C.c_int, [_P, C.c_longlong, C.c_longlong, C.POINTER(C.c_uint32)]),
    "xm_debug_event_outputs": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "xm_stage_rectify": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P]),
    "xm_stage_rectify_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P, _P]),
    "xm_stage_point_cloud": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P]),
    "xm_stage_event_disparity": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_int, _P, _P]),
    "xm_stage_disp_map_projector_view": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P]),
    "xm_stage_disp_map_camera_view": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P]),
    "xm_stage_remap_rectified_disp_map_to_proj": (C.c_int, [_P, _P, _P]),
    "xm_stage_disparity_to_depth": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "xm_stage_colorize_depth_from_disp": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "xm_shard_minmax": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, _P]),
    "xm_shard_minmax_device": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, _P]),
    "xm_shard_scatter_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int, C.c_uint64, _P, C.c_uint32, _P]),
    "xm_shard_clear": (C.c_int, [_P, _P]),
    "xm_shard_scatter": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int, C.c_uint64, _P, C.c_uint32, _P]),
    "xm_shard_finish": (C.c_int, [_P, _P, C.c_uint32, _P, _P]),
    "xm_shard_decode_u16": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, _P]),
    "xm_shard_finish_u16": (C.c_int, [_P, _P, _P, _P]),
    "xm_shard_finish_u16_band": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "xm_k2_patch_cols_max": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "xm_frame_event_filter": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t, _P, C.c_int, C.c_int, _P, C.POINTER(C.c_size_t)]),
    "xm_find_pauses": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, C.c_int64, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "xm_ingest_create": (C.c_int, [_P, C.POINTER(xm_ingest_config), C.POINTER(_P)]),
    "xm_ingest_destroy": (None, [_P]),
    "xm_ingest_push": (C.c_int, [_P, _P, C.c_size_t]),
    "xm_ingest_push_pinned": (C.c_int, [_P, _P, C.c_size_t]),
    "xm_ingest_poll": (C.c_int, [_P, C.POINTER(xm_ingest_frame)]),
    "xm_ingest_poll_owned": (C.c_int, [_P, C.POINTER(xm_ingest_frame), C.POINTER(_P)]),
    "xm_frame_pool_release": (None, [_P, _P, C.c_int]),
    "xm_frame_pool_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "xm_ingest_backlog": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64)]),
    "xm_ingest_flush": (C.c_int, [_P]),
    "xm_ingest_frame_valid": (C.c_int, [_P, C.c_uint64]),
    "xm_ingest_reset": (C.c_int, [_P]),
    "xm_shard_cols_info": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "xm_shard_cols_pack": (C.c_int, [_P, _P, _P, _P, C.c_size_t, _P, C.c_size_t]),
    "xm_shard_cols_scatter": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_uint64, _P, C.c_size_t, C.c_int, C.c_int, C.c_size_t, _P]),
    "xm_shard_cols_failed": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "xm_shard_cols_last_k1_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "xm_create_sharded": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(xm_config), C.POINTER(_P)]),
    "xm_sharded_process_frame": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int, _P, _P, C.POINTER(xm_frame_stats)]),
    "xm_sharded_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "xm_sharded_destroy": (None, [_P]),
    "xm_sharded_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "xm_shard_comm_id": (C.c_int, [_P]),
    "xm_shard_comm_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_uint64, C.POINTER(_P)]),
    "xm_shard_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "xm_shard_comm_frame": (C.c_int, [_P, _P, _P, _P, C.c_size_t, _P, _P]),
    "xm_shard_comm_frame_keys": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_int, C.c_uint64, _P, _P]),
    "xm_shard_comm_failed": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "xm_shard_comm_destroy": (None, [_P]),
    "xm_ingest_device_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "xm_ingest_host_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "xm_activity_create": (C.c_int, [_P, C.c_int64, C.c_size_t, C.POINTER(_P)]),
    "xm_activity_destroy": (None, [_P]),
    "xm_activity_process": (C.c_int, [_P, _P, C.c_size_t, _P, C.POINTER(C.c_size_t)]),
    "xm_activity_reset": (C.c_int, [_P]),
    "xm_activity_set_rule": (C.c_int, [_P, C.c_int]),
    "xm_activity_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "xm_ingest_activity_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "xm_ingest_fused_first_passes": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "xm_evt3_create": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]),
    "xm_evt3_wait_for_time_base": (C.c_int, [_P, C.c_int]),
    "xm_evt3_destroy": (None, [_P]),
    "xm_evt3_reset": (C.c_int, [_P]),
    "xm_evt3_decode": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "xm_ingest_push_evt3": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]),
    "xm_evt2_create": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]),
    "xm_evt2_decode": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "xm_ingest_push_evt2": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]),
    "xm_eval_stats": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(xm_eval_result)]),
    "xm_build_x_map": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "xm_stream": (_P, [_P, C.c_int]),
    "xm_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "xm_host_free": (C.c_int, [_P, _P]),
    "xm_dev_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "xm_dev_free": (C.c_int, [_P, _P]),
    "xm_dev_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "xm_dev_download": (C.c_int, [_P, _P, _P, C.c_size_t]),
}

_lib = None


def load_library(build_if_missing: bool = True) -> C.CDLL:
    """dlopen libxmaps_hip.so (building it first when needed).  Raises -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    alt = os.environ.get("XM_LIB")  # experiments: an alternative build of the library (e.g. an ablation / prototype build)
    if alt:
        lib = C.CDLL(os.path.abspath(alt))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib
    if build_if_missing and needs_build():
        if _hipcc() is not None:
            build_native()
        elif not os.path.exists(LIB_PATH):
            raise XMapsNativeError(f"{LIB_PATH} is missing and hipcc is not available to build it")
    if not os.path.exists(LIB_PATH):
        raise XMapsNativeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return (load_library().xm_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, *, index_error_ok: bool = False) -> int:
    """Map C-ABI error codes onto the exceptions the reference's NumPy code raises."""
    if rc == XM_OK:
        return rc
    msg = last_error()
    if rc == XM_ERR_INDEX:
        if index_error_ok:
            return rc
        raise IndexError(msg)  # NumPy fancy indexing out of range
    if rc == XM_ERR_TOO_MANY:
        raise XMapsTooMany(msg)
    if rc == XM_ERR_INVALID or rc == XM_ERR_UNSORTED:
        raise ValueError(msg)
    if rc == XM_ERR_NOMEM:
        raise MemoryError(msg)
    raise XMapsNativeError(f"libxmaps_hip error {rc}: {msg}")


def debug_option(name, value=None):
    """Variant switch for tests / experiments (xm_debug_option): value None removes it, name None removes all."""
    lib = load_library()
    lib.xm_debug_option(None if name is None else name.encode(), None if value is None else str(value).encode())
