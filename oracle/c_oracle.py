"""ctypes face of oracle/xmaps_oracle.c (test infrastructure: checker + CPU baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class xmo_tables(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("cam_w", "cam_h", "proj_w", "proj_h", "rect_w", "rect_h", "xmap_w", "xmap_h",
                                       "x_offset", "camera_view")] + \
               [("p03", C.c_double), ("z_near", C.c_float), ("z_far", C.c_float),
                ("mapx", C.c_void_p), ("mapy", C.c_void_p), ("xmap", C.c_void_p), ("pmapxy", C.c_void_p),
                ("turbo_bgr", C.c_void_p)]


def physical_cores() -> int:
    """cores, not SMT threads (Linux topology files; falls back to half of os.cpu_count())"""
    seen = set()
    try:
        for d in os.listdir("/sys/devices/system/cpu"):
            if d.startswith("cpu") and d[3:].isdigit():
                with open(f"/sys/devices/system/cpu/{d}/topology/thread_siblings_list") as f:
                    seen.add(f.read().strip())
    except OSError:
        seen = set()
    return len(seen) or max(1, (os.cpu_count() or 2) // 2)


def load(omp=False):
    if omp:  # (read by libgomp when it is loaded: threads stay on their cores, one per core)
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")
    name = "libxmaps_oracle_omp.so" if omp else "libxmaps_oracle.so"
    path = os.path.join(HERE, name)
    src = os.path.join(HERE, "xmaps_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, name], check=True, capture_output=True)
    lib = C.CDLL(path)
    lib.xmo_process_frame.restype = C.c_int
    lib.xmo_num_threads.restype = C.c_int
    lib.xmo_set_num_threads.argtypes = [C.c_int]
    lib.xmo_set_num_threads.restype = None
    return lib


class COracle:
    def __init__(self, tables, camera_perspective=False, omp=False, threads=0, reuse_outputs=False):
        """threads (omp only): 0 = one per physical core; reuse_outputs: process_ev_frame hands out the SAME output arrays every
        call (the timed baseline: no 20 MB of fresh pages per frame)"""
        from xmaps_oracle import _turbo_bgr
        self.lib = load(omp)
        if omp:
            self.lib.xmo_set_num_threads(int(threads) if threads else physical_cores())
        self._reuse = {} if reuse_outputs else None
        self.keep = [np.ascontiguousarray(tables[k], dtype=np.int16) for k in
                     ("cam_mapx_i16", "cam_mapy_i16", "proj_x_map", "disp_proj_mapxy_i16")]
        self.turbo = np.ascontiguousarray(_turbo_bgr())
        t = xmo_tables()
        t.cam_h, t.cam_w = self.keep[0].shape
        t.proj_h, t.proj_w = self.keep[3].shape[:2]
        t.rect_w, t.rect_h = tables["rect_w"], tables["rect_h"]
        t.xmap_h, t.xmap_w = self.keep[2].shape
        t.x_offset = tables.get("x_offset", 4242)
        t.camera_view = int(camera_perspective)
        t.p03, t.z_near, t.z_far = tables["p03"], tables["z_near"], tables["z_far"]
        t.mapx, t.mapy, t.xmap, t.pmapxy = (a.ctypes.data for a in self.keep)
        t.turbo_bgr = self.turbo.ctypes.data
        self.t = t
        self.camera = camera_perspective
        self.fshape = (t.cam_h, t.cam_w) if camera_perspective else (t.rect_h, t.rect_w)
        self.oshape = (t.cam_h, t.cam_w) if camera_perspective else (t.proj_h, t.proj_w)
        self.key = np.zeros(self.fshape, np.uint64)
        self.threads = self.lib.xmo_num_threads()

    def process_ev_frame(self, x, y, t, want_events=True):
        x = np.ascontiguousarray(x, np.uint16)
        y = np.ascontiguousarray(y, np.uint16)
        t = np.ascontiguousarray(t, np.int64)
        n = len(t)
        if self._reuse is not None and self._reuse:
            out = dict(self._reuse)
        else:
            out = {"disp_map": np.empty(self.fshape, np.float32), "depth": np.empty(self.oshape, np.float32),
                   "bgr": np.empty(self.oshape + (3,), np.uint8)}
            if self._reuse is not None:
                self._reuse = dict(out)
        mask = np.empty(n, np.uint8) if want_events else None
        dev = np.empty(n, np.int16) if want_events else None
        ninl = C.c_int64(0)
        p = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(None)
        rc = self.lib.xmo_process_frame(C.byref(self.t), p(x), p(y), p(t), C.c_int64(n), p(self.key), p(out["disp_map"]),
                                        p(out["depth"]), p(out["bgr"]), p(mask), p(dev), C.byref(ninl))
        if rc != 0:
            raise IndexError("event indexed outside a table/frame")
        out["n_inliers"] = ninl.value
        if want_events:
            out["mask"] = mask.astype(bool)
            out["disp"] = dev[out["mask"]]
        return out
