#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN FUNCTIONS.

Run only in the build container (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference modules import numba / cv2 / metavision at module top; none of them is installed
offline, so they are replaced by in-process stubs before the import:
numba.jit -> identity, prange -> range, cv2 -> the 4 constants read at import time, metavision ->
dummy classes.  Under these stubs every function captured below runs UNMODIFIED (NumPy only, or a
Numba body executed as plain Python).  Nothing from /root/reference is copied: the .npz files hold
inputs and the outputs the reference produced for them -- data, not source.

Not capturable here (they call into OpenCV): cv2.dilate / cv2.remap (A4), cv2.applyColorMap (A7),
CamProjMaps.__post_init__ (stereoRectify & friends).  Those stay "parity unpinned".
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/python"


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m


def _jit(*a, **k):
    return a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)


EventCD = np.dtype({"names": ["x", "y", "p", "t"], "formats": ["<u2", "<u2", "<i2", "<i8"],
                    "offsets": [0, 2, 4, 8], "itemsize": 16})


class _Alg:
    def __init__(self, *a, **k):
        pass

    get_empty_output_buffer = staticmethod(lambda: None)


def import_reference():
    _stub("numba", jit=_jit, njit=_jit, prange=range)
    _stub("cv2", BORDER_REPLICATE=1, BORDER_CONSTANT=0, INTER_NEAREST=0, COLORMAP_TURBO=20)
    _stub("metavision_sdk_base", EventCD=EventCD, EventCDBuffer=object)
    _stub("metavision_sdk_core", PolarityFilterAlgorithm=_Alg)
    _stub("metavision_sdk_cv", ActivityNoiseFilterAlgorithm=_Alg)
    _stub("metavision_sdk_ui", BaseWindow=object, MTWindow=object, UIAction=object, UIKeyEvent=object,
          EventLoop=object)
    sys.path.insert(0, REF)
    import cam_proj_calibration  # noqa
    import disp_to_depth  # noqa
    import frame_event_filter  # noqa
    import proj_time_map  # noqa
    import trigger_finder  # noqa
    import x_map  # noqa
    import x_maps_disparity  # noqa
    return types.SimpleNamespace(calib=cam_proj_calibration, d2d=disp_to_depth, fef=frame_event_filter,
                                 ptm=proj_time_map, tf=trigger_finder, xmap=x_map, xmd=x_maps_disparity)


def _duck_maps(mapx, mapy, rect_h, rect_w, cam_h, cam_w):
    calib = types.SimpleNamespace(rect_image_height=rect_h, rect_image_width=rect_w,
                                  camera_height=cam_h, camera_width=cam_w)
    return types.SimpleNamespace(disp_cam_mapx_i16=mapx, disp_cam_mapy_i16=mapy, calib=calib)


def run_event_path(ref, tables, events, t_px_scale, x_offset=4242):
    """A1 -> A2 -> A3 + A3' through the reference's functions. `events` = dict-like with x, y, t."""
    CPM = ref.calib.CamProjMaps
    obj = _duck_maps(tables["mapx"], tables["mapy"], tables["rect_h"], tables["rect_w"],
                     tables["cam_h"], tables["cam_w"])
    xr, yr = CPM.rectify_cam_coords_i16(obj, events)
    with np.errstate(all="ignore"):
        disp, mask = ref.xmd.compute_disparity(xr, yr, events["t"], tables["xmap"], t_px_scale, x_offset)
    out = {"xr": xr, "yr": yr, "disp": disp, "mask": mask}
    try:
        out["disp_map_proj"] = CPM.compute_disp_map_projector_view(obj, xr, yr, mask, disp)
    except IndexError:
        out["proj_index_error"] = np.array(1)
    out["disp_map_cam"] = CPM.compute_disp_map_camera_view(obj, events, mask, disp)
    return out


def small_tables(rng, cam_w, cam_h, rect_w, rect_h, xmap_w, x_offset=4242):
    ys, xs = np.mgrid[0:cam_h, 0:cam_w].astype(np.float64)
    sx, sy = rect_w / cam_w, rect_h / cam_h
    mapx = np.rint(0.72 * sx * xs + 0.1 * rect_w * 0.5 + 0.05 * ys).astype(np.int16)
    mapy = np.rint(1.04 * sy * ys - 0.02 * rect_h + 0.02 * xs).astype(np.int16)  # leaves the frame at both ends
    yr, tc = np.mgrid[0:rect_h, 0:xmap_w].astype(np.float64)
    xmap = np.rint(x_offset + 0.15 * rect_w + tc * (0.8 * rect_w) / xmap_w + 0.02 * yr).astype(np.int16)
    xmap[:, 0] = 0
    xmap[rng.random(xmap.shape) < 0.03] = 0  # undefined cells
    return {"mapx": mapx, "mapy": mapy, "xmap": xmap, "rect_h": rect_h, "rect_w": rect_w,
            "cam_h": cam_h, "cam_w": cam_w}


def events_struct(x, y, t, p=None):
    ev = np.zeros(len(x), dtype=EventCD)
    ev["x"], ev["y"], ev["t"] = x, y, t
    ev["p"] = 1 if p is None else p
    return ev


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def g1_event_path(ref):
    # ---- G1a / G1b: seeded synthetic frames, sorted time, scan-correlated x
    for tag, n, (cw, ch), seed in (("g1a_n1000", 1000, (64, 48), 11), ("g1b_n100000", 100_000, (160, 120), 12)):
        rng = np.random.default_rng(seed)
        rw, rh = round(cw * 2.75), round(ch * 2.75)
        xw = cw
        tb = small_tables(rng, cw, ch, rw, rh, xw)
        t_rel = np.sort(rng.integers(0, 13_000, n))
        x = np.clip(np.rint(t_rel / 13_000 * cw + rng.normal(0, 2.0, n)), 0, cw - 1).astype(np.uint16)
        y = rng.integers(0, ch, n).astype(np.uint16)
        ev = events_struct(x, y, 5_000_000 + t_rel)
        out = run_event_path(ref, tb, ev, xw - 1)
        save(f"{tag}.npz", x=ev["x"], y=ev["y"], t=ev["t"], t_px_scale=np.array(xw - 1), **tb, **out)

    # ---- G1c: unsorted t + i.i.d. pixels (raster/filtered-order case), heavy duplicates with differing disparity
    rng = np.random.default_rng(13)
    cw, ch = 32, 24
    rw, rh = 88, 66
    tb = small_tables(rng, cw, ch, rw, rh, 32)
    n = 5000
    ev = events_struct(rng.integers(0, cw, n).astype(np.uint16), rng.integers(0, ch, n).astype(np.uint16),
                       1_000 + rng.integers(0, 9_000, n))
    out = run_event_path(ref, tb, ev, 31)
    save("g1c_unsorted_dups.npz", x=ev["x"], y=ev["y"], t=ev["t"], t_px_scale=np.array(31), **tb, **out)

    # ---- G1d: exact rint ties. S = 64 (X-map width 65), tmax - tmin = 128, t = tmin + 2k + 1 -> tn*S = k + 0.5
    rng = np.random.default_rng(14)
    tb = small_tables(rng, 32, 24, 88, 66, 65)
    k = np.arange(0, 64)
    t = np.concatenate(([7_000], 7_000 + 2 * k + 1, [7_128])).astype(np.int64)
    n = len(t)
    ev = events_struct(rng.integers(0, 32, n).astype(np.uint16), rng.integers(2, 22, n).astype(np.uint16), t)
    out = run_event_path(ref, tb, ev, 64)
    save("g1d_rint_ties.npz", x=ev["x"], y=ev["y"], t=ev["t"], t_px_scale=np.array(64), **tb, **out)

    # ---- G1e: boundary rows yr in {-1, 0, H-2, H-1}, disp == 0, disp == -1, undefined cell, negative-column wrap
    cw, ch, rw, rh, xw = 8, 8, 40, 30, 16
    mapx = np.zeros((ch, cw), np.int16)
    mapy = np.zeros((ch, cw), np.int16)
    xmap = np.zeros((rh, xw), np.int16)
    rows = [-1, 0, rh - 2, rh - 1, 5, 6, 7, 9]
    for yy in range(ch):
        mapy[yy, :] = rows[yy]
        mapx[yy, :] = np.arange(cw) * 3 + 2
    xmap[:, :] = 4242 + 10 + np.arange(xw)[None, :] * 2
    xmap[5, :] = 4242 + (np.arange(xw) % cw) * 3 + 2  # row 5: disp == 0 when column == x (mod 8)
    xmap[6, :] = 4242 + (np.arange(xw) % cw) * 3 + 1  # row 6: disp == -1 -> rejected
    xmap[7, :] = 0                                     # row 7: undefined
    mapx[7, :] = -20 + np.arange(cw)                   # row 9 (cam y = 7): negative xr ...
    xmap[9, :] = 4242 - 12                             # ... and xp - 4242 = -12 -> column wraps to rw - 12
    ys, xs = np.mgrid[0:ch, 0:cw]
    x = np.tile(xs.ravel(), 2).astype(np.uint16)
    y = np.tile(ys.ravel(), 2).astype(np.uint16)
    t = (100 + np.arange(len(x)) * 3).astype(np.int64)
    tb = {"mapx": mapx, "mapy": mapy, "xmap": xmap, "rect_h": rh, "rect_w": rw, "cam_h": ch, "cam_w": cw}
    ev = events_struct(x, y, t)
    out = run_event_path(ref, tb, ev, xw - 1)
    save("g1e_edges.npz", x=ev["x"], y=ev["y"], t=ev["t"], t_px_scale=np.array(xw - 1), **tb, **out)

    # ---- G1f / G1g: float t (eval caller: already-normalised time surface values), f32 and f64
    rng = np.random.default_rng(15)
    tb = small_tables(rng, 64, 48, 176, 132, 64)
    for tag, dt in (("g1f_float32_t", np.float32), ("g1g_float64_t", np.float64)):
        surf = rng.random((48, 64)).astype(dt)
        surf[rng.random(surf.shape) < 0.3] = 0
        yy, xx = np.nonzero(surf > 0)
        events = {"x": xx, "y": yy, "t": surf[surf > 0]}
        out = run_event_path(ref, tb, events, 63)
        save(f"{tag}.npz", x=xx, y=yy, t=events["t"], t_px_scale=np.array(63), **tb, **out)

    # ---- G1h: all timestamps equal (0/0 -> NaN -> int16 cast; pins what NumPy does here on x86-64)
    rng = np.random.default_rng(16)
    tb = small_tables(rng, 32, 24, 88, 66, 32)
    tb["xmap"][:, 0] = tb["xmap"][:, 1]  # make column 0 defined so the outcome is visible
    n = 200
    ev = events_struct(rng.integers(0, 32, n).astype(np.uint16), rng.integers(0, 24, n).astype(np.uint16),
                       np.full(n, 4_242_424))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = run_event_path(ref, tb, ev, 31)
    save("g1h_equal_t.npz", x=ev["x"], y=ev["y"], t=ev["t"], t_px_scale=np.array(31), **tb, **out)


def g2_frame_stages(ref):
    rng = np.random.default_rng(21)
    disp = rng.integers(0, 400, (32, 32)).astype(np.float32)
    disp[rng.random(disp.shape) < 0.4] = 0
    disp[0, :4] = [1, 2, 3, 100000]
    P = np.zeros((3, 4))
    P[0, 3] = 193.4075
    depth = ref.d2d.disparity_to_depth_rectified(disp, P)
    u8 = ref.d2d.clip_normalize_uint8_depth_frame(depth, min_value=0.1, max_value=1.2)
    frame = rng.integers(0, 255, (32, 32, 3)).astype(np.uint8)
    white = ref.d2d.apply_white_mask(frame.copy(), u8)
    Pn = P.copy()
    Pn[0, 3] = -50.0
    depth_neg = ref.d2d.disparity_to_depth_rectified(disp, Pn)
    save("g2_frame_stages.npz", disp=disp, p03=np.array(P[0, 3]), depth=depth, z_near=np.array(0.1),
         z_far=np.array(1.2), u8=u8, frame_in=frame, frame_white=white, depth_neg=depth_neg,
         p03_neg=np.array(-50.0))


def g3_x_map(ref):
    rng = np.random.default_rng(31)
    h, w, tw = 48, 64, 32
    tm = ref.ptm.generate_linear_projector_time_map(w, h, True)
    tm = (tm + rng.normal(0, 2e-4, tm.shape)).astype(np.float32)
    tm[:3, :] = 0
    tm[-2:, :] = 0
    tm[:, :5] = 0
    tm[rng.random(tm.shape) < 0.05] = 0
    tm[10, 20] = tm[10, 21]  # equal candidates -> first minimum must win
    x_map, t_diffs = ref.xmap.compute_x_map_from_time_map(tm, tw, tw - 1, 4242, w)
    save("g3_x_map.npz", time_map=tm, x_map_width=np.array(tw), t_px_scale=np.array(tw - 1),
         num_scanlines=np.array(w), x_map=x_map, t_diffs=t_diffs)


def g4_time_map(ref):
    out = {}
    for (w, h) in ((7, 5), (16, 9)):
        for up in (True, False):
            out[f"tm_{w}x{h}_{'up' if up else 'down'}"] = ref.ptm.generate_linear_projector_time_map(w, h, up)
    save("g4_time_map.npz", **out)


class _Buf:
    """Stand-in for a Metavision EventCDBuffer: .numpy() returns the structured array."""

    def __init__(self, a):
        self._a = a

    def numpy(self):
        return self._a


class _Pool:
    def return_buf(self, b):
        pass

    def get_buf(self):
        return None


class _Stats:
    def __init__(self):
        self.counts = {}

    def count(self, k, n=1):
        self.counts[k] = self.counts.get(k, 0) + n

    def add_metric(self, *a):
        pass

    def measure_time(self, k):
        import contextlib
        return contextlib.nullcontext()


def g5_trigger_and_filters(ref):
    # stream of 6 projector frames at 60 fps: 13 ms scan + ~3.6 ms silent gap, jittered so that the
    # packet grid is not phase-locked to the frames; one in-scan pause of 45 us to create a false pause
    rng = np.random.default_rng(51)
    fps = 60
    chunks = []
    t0 = 1_000_000
    for f in range(8):
        n = 3000 + int(rng.integers(0, 500))
        # frame period 16.6 ms < 1e6/fps so consecutive frame-end events satisfy the `<= 1e6/fps` gate
        start = t0 + f * 16_600 + int(rng.integers(0, 50))
        tt = np.sort(rng.integers(0, 13_000, n)) + start
        # densify so that in-scan gaps stay < 40 us
        tt = np.unique(np.concatenate((tt, np.arange(start, start + 13_000, 25))))
        chunks.append(tt)
    t = np.concatenate(chunks).astype(np.int64)
    n = len(t)
    ev = events_struct(rng.integers(0, 64, n).astype(np.uint16), rng.integers(0, 48, n).astype(np.uint16), t)
    frames = []
    stats = _Stats()
    tf = ref.tf.RobustTriggerFinder(projector_fps=fps, stats=stats, pool=_Pool(),
                                    frame_callback=lambda e: frames.append(e.copy()))
    packet_us = int(1e6 / fps / 4)
    edges = np.arange(t[0], t[-1] + packet_us, packet_us)
    cuts = np.searchsorted(t, edges)
    for a, b in zip(cuts[:-1], cuts[1:]):
        tf.process_events(_Buf(ev[a:b]))
    assert len(frames) >= 2, len(frames)
    save("g5_trigger.npz", x=ev["x"], y=ev["y"], t=ev["t"], fps=np.array(fps), packet_cuts=cuts,
         n_frames=np.array(len(frames)),
         frame_first_t=np.array([f["t"][0] for f in frames]), frame_last_t=np.array([f["t"][-1] for f in frames]),
         frame_len=np.array([len(f) for f in frames]),
         trig_ok=np.array(stats.counts.get("trig ✅", 0)), trig_fail=np.array(stats.counts.get("trig ❌", 0)))

    # frame filters on one small frame (x-proj stand-in = a seeded int16 array, as the pipe passes xr)
    rng = np.random.default_rng(52)
    n = 3000
    fe = events_struct(rng.integers(0, 40, n).astype(np.uint16), rng.integers(0, 30, n).astype(np.uint16),
                       np.sort(rng.integers(10, 12_000, n)).astype(np.int64),
                       p=(rng.random(n) < 0.9).astype(np.int16))
    xp = rng.integers(0, 90, int((fe["p"] == 1).sum())).astype(np.int16)
    out = {}
    for cls in ("LastEventPerXYFilter", "FirstEventPerXYFilter", "MeanFirstLastEventPerXYFilter", "FirstEventPerYTFilter"):
        r = getattr(ref.fef, cls)().filter_events(fe, xp)
        out[f"{cls}_x"], out[f"{cls}_y"], out[f"{cls}_t"], out[f"{cls}_p"] = r["x"], r["y"], r["t"], r["p"]
    save("g5_filters.npz", x=fe["x"], y=fe["y"], t=fe["t"], p=fe["p"], xp=xp, **out)


def g8_eval_metrics():
    """The evaluation metrics of python/eval/create_evaluation_table.py:14-63 (class evaluation_stats, load_and_filter) run on
    synthetic depth maps in centimetres (the script's unit: min 20, max 120).  The module imports esl_utilities (which pulls
    cv2 / matplotlib) at its top: stubbed -- the two functions captured here do not touch it."""
    _stub("esl_utilities", utils=object)
    sys.path.insert(0, os.path.join(REF, "eval"))
    import create_evaluation_table as cet
    rng = np.random.default_rng(88)
    cases = {}
    for name, (h, w, hole_gt, hole_est, noise) in {"a": (48, 64, 0.2, 0.3, 0.4), "b": (120, 160, 0.35, 0.25, 2.5),
                                                      "c": (31, 47, 0.0, 0.0, 8.0)}.items():
        yy, xx = np.mgrid[0:h, 0:w]
        gt = (60 + 25 * np.sin(xx / 37.0) + 15 * np.cos(yy / 23.0)).astype(np.float32)
        gt[rng.random(gt.shape) < hole_gt] = 0
        est_raw = (gt + rng.normal(0, noise, gt.shape)).astype(np.float32)
        est_raw[rng.random(gt.shape) < hole_est] = 0
        est_raw[rng.random(gt.shape) < 0.02] = 150  # beyond max_depth
        est_raw[rng.random(gt.shape) < 0.02] = 5    # below min_depth
        path = os.path.join("/tmp", f"g8_{name}.npy")
        np.save(path, est_raw)
        est = cet.load_and_filter(path, gt, 20, 120)
        st = cet.evaluation_stats(est, gt)
        cases[name] = dict(gt=gt, est_raw=est_raw, est=est,
                           res=np.array([st.fillrate, st.rmse, st.perc_1, st.perc_5, st.perc_10, st.margin], np.float64))
    # degenerate: estimate empty -> rmse branch "no valid values" (0)
    gt = cases["a"]["gt"]
    st = cet.evaluation_stats(np.zeros_like(gt), gt)
    save("g8_eval_metrics.npz", min_depth=np.array(20.0), max_depth=np.array(120.0),
         empty_res=np.array([st.fillrate, st.rmse, st.perc_1, st.perc_5, st.perc_10, st.margin], np.float64),
         **{f"{k}_{f}": v[f] for k, v in cases.items() for f in v})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        g8_eval_metrics()
        sys.exit(0)
    assert os.path.isdir(REF), "needs /root/reference (build container only)"
    ref = import_reference()
    g1_event_path(ref)
    g2_frame_stages(ref)
    g3_x_map(ref)
    g4_time_map(ref)
    g5_trigger_and_filters(ref)


def g6_calibration_data():
    """The numbers of the reference's calibration file data/ESL_calib_hhi.yaml (intrinsics, distortion, relative pose and
    the R1/R2/P1/P2/Q that OpenCV's stereoRectify wrote into it) as a fixture: data, so that GPU-box tests can build
    realistic tables without /root/reference."""
    import yaml
    with open("/root/reference/data/ESL_calib_hhi.yaml") as f:
        d = yaml.safe_load(f)

    def mat(name):
        n = d[name]
        return np.array(n["data"], dtype=np.float64).reshape(n["rows"], n["cols"])

    save("g6_esl_calib.npz", camera_K=mat("camera_intrinsic_matrix"), camera_D=mat("camera_distortion_coefficients"),
         projector_K=mat("projector_intrinsic_matrix"), projector_D=mat("projector_distortion_coefficients"),
         R=mat("relative_rotation"), T=mat("relative_translation"), R1=mat("R1"), R2=mat("R2"), P1=mat("P1"), P2=mat("P2"),
         Q=mat("Q"))


if __name__ == "__main__":
    g6_calibration_data()


def g7_eval_caller(ref):
    """The offline-evaluation caller (python/eval/compute_depth_x_maps.py:81-114) through the reference's functions:
    time surface -> normalise -> events (raster order, float t) -> rectify i16 / f32 -> compute_event_disparity ->
    compute_disp_map_camera_view -> disparity_to_depth_rectified -> construct_point_cloud."""
    rng = np.random.default_rng(71)
    cw, ch, rw, rh, xw = 64, 48, 176, 132, 64
    tb = small_tables(rng, cw, ch, rw, rh, xw)
    # f32 rectify maps (disp_cam_map{x,y}_f32) whose rint gives the i16 LUT
    mapx_f = (tb["mapx"] + rng.uniform(-0.45, 0.45, tb["mapx"].shape)).astype(np.float32)
    mapy_f = (tb["mapy"] + rng.uniform(-0.45, 0.45, tb["mapy"].shape)).astype(np.float32)
    cam_image = (rng.random((ch, cw)) * 0.8 + 0.1).astype(np.float64)
    cam_image[rng.random(cam_image.shape) < 0.35] = 0
    raw = cam_image.copy()
    # lines 83-87 of the script
    cam_image = (cam_image - np.min(cam_image[cam_image != 0])) / (np.max(cam_image[cam_image != 0]) - np.min(cam_image[cam_image != 0]))
    cam_image[cam_image < 0] = 0
    event_y = np.argwhere(cam_image > 0)[:, 0]
    event_x = np.argwhere(cam_image > 0)[:, 1]
    event_t = cam_image[cam_image > 0]
    events = {"x": event_x, "y": event_y, "t": event_t}
    CPM = ref.calib.CamProjMaps
    obj = _duck_maps(tb["mapx"], tb["mapy"], rh, rw, ch, cw)
    obj.disp_cam_mapx_f32, obj.disp_cam_mapy_f32 = mapx_f, mapy_f
    Q = np.array([[1, 0, 0, -80.5], [0, 1, 0, -60.25], [0, 0, 0, 540.0], [0, 0, -7.75, 0]], dtype=np.float64)
    obj.Q = Q
    xr_f, yr_f = CPM.rectify_cam_coords_f32(obj, events)
    xr, yr = CPM.rectify_cam_coords_i16(obj, events)
    disp, mask = ref.xmd.compute_disparity(xr, yr, events["t"], tb["xmap"], xw - 1, 4242)
    disp_map = CPM.compute_disp_map_camera_view(obj, events, mask, disp)
    P = np.zeros((3, 4))
    P[0, 3] = 69.4
    depth = ref.d2d.disparity_to_depth_rectified(disp_map, P)
    cloud = CPM.construct_point_cloud(obj, xr_f[mask], yr_f[mask], disp)
    save("g7_eval_caller.npz", raw_time_surface=raw, mapx=tb["mapx"], mapy=tb["mapy"], mapx_f32=mapx_f, mapy_f32=mapy_f,
         xmap=tb["xmap"], rect_h=np.array(rh), rect_w=np.array(rw), t_px_scale=np.array(xw - 1), Q=Q, p03=np.array(69.4),
         event_x=event_x, event_y=event_y, event_t=event_t, disp=disp, mask=mask, disp_map=disp_map, depth=depth,
         xr_f32=xr_f, yr_f32=yr_f, cloud=cloud)


if __name__ == "__main__":
    g7_eval_caller(import_reference())
