"""ESL-like synthetic rig: the reference's real camera/projector geometry + a rendered 3-D scene.

Stand-in for BASELINE configs 1 and 3 (ESL static seq1), whose RAW recording is not available offline:
intrinsics / relative pose are the numbers of the reference's data/ESL_calib_hhi.yaml (fixture tests/golden/g6_esl_calib.npz),
the camera's lens distortion is the mild one of the live-setup file (data/nebra_evk3.0/X-maps_calibration_8_5mm.yaml:16-26),
the projector is 1080 x 1920 drawing x-slow / y-fast at 60 Hz.  Tables come from x_maps_amd.calibration.build_tables (cv2-free);
events are rendered by intersecting every (sub-sampled) projector ray with a piecewise-planar scene, projecting into the
camera and stamping the pixel with the projector's drawing time in microseconds.  Ground truth = the point's Z in the
rectified frame, which is what depth = P2[0,3] / disparity estimates (python/disp_to_depth.py:46-63).
"""
from __future__ import annotations

import os

import numpy as np

from . import calibration as C
from .synthetic import EVENT_CD_DTYPE

NEBRA_CAMERA_D = np.array([-6.44787441e-04, 5.94768864e-03, -9.99388025e-05, 4.31726843e-04, 1.23020685e-01])


def esl_like_params(calib_npz: str, proj_w=1080, proj_h=1920, cam_w=640, cam_h=480) -> C.CamProjCalibrationParams:
    g = np.load(calib_npz)
    return C.CamProjCalibrationParams(
        cam_w, cam_h, proj_w, proj_h, round(cam_w * 2.75), round(cam_h * 2.75),
        g["camera_K"], NEBRA_CAMERA_D.copy(), g["projector_K"], np.zeros(5), g["R"], g["T"])


def scene_depth(xn, yn):
    """Piecewise-planar scene in the projector frame: tilted back plane + a nearer box + a slanted 'book'.
    Returns Z for the projector ray with normalised coordinates (xn, yn)."""
    z = 0.62 / (1.0 - 0.15 * xn + 0.10 * yn)                    # plane z = 0.62 + 0.15 X - 0.10 Y
    box = (np.abs(xn - 0.02) < 0.05) & (np.abs(yn + 0.08) < 0.07)
    z = np.where(box, 0.50 + 0.0 * xn, z)
    book = (np.abs(xn + 0.09) < 0.045) & (np.abs(yn - 0.12) < 0.09)
    z = np.where(book, 0.55 / (1.0 + 0.6 * xn), z)
    return z


def render_events(cp: C.CamProjCalibrationParams, tables: dict, row_stride=13, t0_us=7_000_000, scan_us=13_000,
                  scan_upwards=True, jitter_us=0.0, seed=0):
    """One projector frame of EventCD records (time-sorted) + per-event ground truth.
    Returns (events, gt) with gt = dict(proj_u, proj_v, z_rect) for the events that landed on the sensor."""
    rng = np.random.default_rng(seed)
    W, H = cp.projector_width, cp.projector_height
    cols = np.arange(W)
    rows = np.arange(rng.integers(0, row_stride), H, row_stride)
    cc, rr = np.meshgrid(cols, rows, indexing="ij")
    cc, rr = cc.ravel(), rr.ravel()
    K = cp.projector_K
    xn, yn = (cc - K[0, 2]) / K[0, 0], (rr - K[1, 2]) / K[1, 1]  # projector distortion is zeroed (calib:87-89)
    z = scene_depth(xn, yn)
    xyz_p = np.stack((xn * z, yn * z, z), -1)
    # x_proj = R x_cam + T  (relative_rotation / relative_translation of the calibration file)
    xyz_c = (xyz_p - cp.cam2proj_T.reshape(1, 3)) @ cp.cam2proj_R  # = R^T (x_p - T)
    uv = C.project_points(xyz_c, cp.camera_K, cp.camera_D)
    u, v = np.rint(uv[:, 0]).astype(np.int64), np.rint(uv[:, 1]).astype(np.int64)
    ok = (xyz_c[:, 2] > 0) & (u >= 0) & (u < cp.camera_width) & (v >= 0) & (v < cp.camera_height)
    rr_t = (H - 1 - rr) if scan_upwards else rr
    tnorm = (cc * H + rr_t) / float(W * H)
    t = t0_us + np.rint(tnorm * scan_us + rng.normal(0.0, jitter_us, len(tnorm)) if jitter_us else tnorm * scan_us).astype(np.int64)
    order = np.argsort(t[ok], kind="stable")
    evs = np.zeros(int(ok.sum()), EVENT_CD_DTYPE)
    evs["x"], evs["y"], evs["t"], evs["p"] = u[ok][order], v[ok][order], t[ok][order], 1
    z_rect = (xyz_c[ok] @ tables["R1"].T)[:, 2][order]  # camera is the pair's geometric first view (R, T map camera -> projector)
    return evs, {"proj_u": cc[ok][order], "proj_v": rr[ok][order], "z_rect": z_rect}


def make_esl_like(calib_npz: str | None = None, device: int = 0, x_map_fn=None, **kw):
    if calib_npz is None:
        calib_npz = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6_esl_calib.npz")
    cp = esl_like_params(calib_npz)
    tables = C.build_tables(cp, device=device, x_map_fn=x_map_fn)
    evs, gt = render_events(cp, tables, **kw)
    return cp, tables, evs, gt


def render_stream(cp: C.CamProjCalibrationParams, tables: dict, n_frames: int, period_us: int = 16_600, scan_us: int = 13_000,
                  row_stride: int = 13, t_start_us: int = 2_000_000, neg_fraction: float = 0.1, gap_noise_every: int = 5,
                  seed: int = 0):
    """BASELINE config 3 stand-in: `n_frames` consecutive ESL-like projector frames as ONE EventCD stream with real-looking
    microsecond stamps -- a ~13 ms scan, then a ~3.6 ms dark gap (period 16.6 ms, 60 Hz) -- plus what a camera adds:
    `neg_fraction` negative-polarity events sprinkled over the scans (the polarity filter removes them, pipe:43,114) and a
    lone positive noise event inside every `gap_noise_every`-th gap (the trigger finder then cuts that frame at the noise
    event, trigger_finder.py:158-172).  Returns (stream, frames) with frames = the rendered per-frame event arrays."""
    rng = np.random.default_rng(seed)
    chunks, frames = [], []
    for f in range(n_frames):
        t0 = t_start_us + f * period_us
        evs, _ = render_events(cp, tables, row_stride=row_stride, t0_us=t0, scan_us=scan_us, seed=seed * 1000 + f)
        frames.append(evs)
        parts = [evs]
        if neg_fraction > 0:
            k = int(len(evs) * neg_fraction)
            neg = np.zeros(k, EVENT_CD_DTYPE)
            neg["t"] = rng.integers(t0, t0 + scan_us, k)
            neg["x"] = rng.integers(0, cp.camera_width, k)
            neg["y"] = rng.integers(0, cp.camera_height, k)
            neg["p"] = 0
            parts.append(neg)
        if gap_noise_every and f % gap_noise_every == gap_noise_every - 1:
            nz = np.zeros(1, EVENT_CD_DTYPE)
            nz["t"] = t0 + scan_us + 1_500
            nz["x"], nz["y"], nz["p"] = rng.integers(0, cp.camera_width), rng.integers(0, cp.camera_height), 1
            parts.append(nz)
        c = np.concatenate(parts)
        chunks.append(c[np.argsort(c["t"], kind="stable")])
    return np.concatenate(chunks), frames
