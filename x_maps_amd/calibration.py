"""Setup-time table builder without OpenCV ("next" row N4 of SURVEY.md section 8(f)).

The reference builds every table the hot path consumes with OpenCV (python/cam_proj_calibration.py:174-270:
cv2.stereoRectify, cv2.initUndistortRectifyMap, cv2.undistortPoints; python/proj_time_map.py:22-29: cv2.remap).
OpenCV is not available offline, so this module restates the same pinhole / Brown-distortion geometry in NumPy,
following OpenCV's documented algorithms (Bouguet rectification as in cv::stereoRectify with CALIB_ZERO_DISPARITY,
alpha < 0).  PARITY STATUS: the rectifying rotations are pinned against the R1 / R2 matrices that the reference's own
calibration files store (data/ESL_calib_hhi.yaml:70-92, written by OpenCV); everything else (new focal length,
principal points, map rounding at exact .5) is checked for geometric self-consistency only (epipolar alignment,
forward/inverse round trip) -- "parity unpinned" against cv2, see DESIGN.md.

Host-side NumPy, runs once per session; the heavy step (X-map from the rectified time map) goes to the GPU
(x_maps_amd.x_map.compute_x_map_from_time_map).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .proj_time_map import generate_linear_projector_time_map


# ---- YAML ------------------------------------------------------------------------------------------------
def read_cv_matrix(data: dict, name: str) -> np.ndarray:
    """OpenCV FileStorage matrix as the calibration app writes it (python/cam_proj_calibration.py:17-28)."""
    node = data.get(name)
    if not isinstance(node, dict) or node.get("type-id") != "opencv_matrix":
        raise ValueError(f"Could not read matrix {name} from calibration data")
    return np.array(node["data"], dtype=np.float64).reshape(node["rows"], node["cols"])


def open_calibration_data(path: str) -> dict:
    import yaml
    with open(path, "r") as f:
        return yaml.safe_load(f)


def read_opencv_filestorage(path: str) -> dict:
    """cv2.FileStorage(path, FILE_STORAGE_READ) for the YAML flavour OpenCV writes (`%YAML:1.0` directive, matrices tagged
    `!!opencv-matrix` with rows / cols / dt / data) -- what the ESL dataset's calibration files are
    (python/cam_proj_calibration.py:119-125 reads cam_K, cam_kc, proj_K, proj_kc, R, T from one).  Returns name -> ndarray
    (matrices) or plain Python values."""
    import yaml

    class _Loader(yaml.SafeLoader):
        pass

    def _matrix(loader, node):
        m = loader.construct_mapping(node, deep=True)
        dt = str(m.get("dt", "d"))
        np_dt = {"d": np.float64, "f": np.float32, "i": np.int32, "u": np.uint8, "s": np.int16, "w": np.uint16, "c": np.int8}[dt[-1]]
        ch = int(dt[:-1]) if len(dt) > 1 else 1
        a = np.array(m["data"], dtype=np_dt)
        return a.reshape(int(m["rows"]), int(m["cols"])) if ch == 1 else a.reshape(int(m["rows"]), int(m["cols"]), ch)

    _Loader.add_constructor("tag:yaml.org,2002:opencv-matrix", _matrix)
    with open(path, "r") as f:
        text = f.read()
    lines = text.splitlines()
    if lines and lines[0].startswith("%YAML"):  # OpenCV writes "%YAML:1.0", which is not a valid YAML directive
        lines = lines[1:]
    return yaml.load("\n".join(lines), Loader=_Loader) or {}


# ---- small geometry kit ----------------------------------------------------------------------------------
def rodrigues(v) -> np.ndarray:
    """rotation vector (3,) -> matrix, or matrix (3,3) -> vector (cv::Rodrigues)."""
    v = np.asarray(v, dtype=np.float64)
    if v.shape == (3, 3):
        u, _, vt = np.linalg.svd(v)
        r = u @ vt
        rv = np.array([r[2, 1] - r[1, 2], r[0, 2] - r[2, 0], r[1, 0] - r[0, 1]])
        s = np.linalg.norm(rv) * 0.5
        c = np.clip((np.trace(r) - 1.0) * 0.5, -1.0, 1.0)
        theta = np.arccos(c)
        if s < 1e-12:
            if c > 0:
                return np.zeros(3)
            t = (r + np.eye(3)) * 0.5  # theta = pi
            axis = np.sqrt(np.maximum(np.diag(t), 0.0))
            if t[0, 1] < 0:
                axis[1] = -axis[1]
            if t[0, 2] < 0:
                axis[2] = -axis[2]
            return axis / np.linalg.norm(axis) * theta
        return rv * (0.5 / s) * theta
    v = v.reshape(3)
    theta = np.linalg.norm(v)
    if theta < 1e-15:
        return np.eye(3)
    k = v / theta
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(theta) * np.eye(3) + (1 - np.cos(theta)) * np.outer(k, k) + np.sin(theta) * kx


def _dist5(d) -> np.ndarray:
    d = np.zeros(5) if d is None else np.asarray(d, dtype=np.float64).ravel()
    out = np.zeros(5)
    out[:min(5, len(d))] = d[:5]
    return out


def distort_normalized(x, y, d):
    """ideal normalised coords -> distorted normalised coords (Brown model, k1 k2 p1 p2 k3)."""
    k1, k2, p1, p2, k3 = _dist5(d)
    r2 = x * x + y * y
    radial = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return xd, yd


def undistort_points(pts, K, D, R=None, P=None, iters: int = 5):
    """cv::undistortPoints: pixel coords (N,2) of the distorted image -> ideal coords, optionally rotated by R and
    re-projected with P (3x3 or 3x4).  Fixed-point iteration like OpenCV's (5 rounds by default)."""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    k1, k2, p1, p2, k3 = _dist5(D)
    x0 = (pts[:, 0] - cx) / fx
    y0 = (pts[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    if np.any(_dist5(D) != 0):
        for _ in range(iters):
            r2 = x * x + y * y
            icdist = 1.0 / (1 + r2 * (k1 + r2 * (k2 + r2 * k3)))
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x = (x0 - dx) * icdist
            y = (y0 - dy) * icdist
    if R is not None:
        xyz = np.stack((x, y, np.ones_like(x)), 0)
        xyz = np.asarray(R, dtype=np.float64) @ xyz
        x, y = xyz[0] / xyz[2], xyz[1] / xyz[2]
    if P is not None:
        P = np.asarray(P, dtype=np.float64)
        x = P[0, 0] * x + P[0, 2]
        y = P[1, 1] * y + P[1, 2]
    return np.stack((x, y), -1)


def project_points(xyz, K, D=None):
    """3-D points in the camera frame (N,3) -> distorted pixel coords."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    x, y = xyz[:, 0] / xyz[:, 2], xyz[:, 1] / xyz[:, 2]
    xd, yd = distort_normalized(x, y, D)
    return np.stack((K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]), -1)


# ---- cv::stereoRectify (Bouguet) -----------------------------------------------------------------------------------
def _f32(a):
    """round to float32 and back: OpenCV keeps these intermediate points as CV_32F"""
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def _inner_outer_rectangles(K, D, R, P, image_size):
    """icvGetRectangles: a 9 x 9 grid over the source image, undistorted + rectified (float32 points); the largest rectangle
    inscribed in / the bounding box of its image.  Returns ((x, y, w, h) inner, (x, y, w, h) outer), float32 arithmetic."""
    nx, ny = image_size
    N = 9
    f = np.float32
    pts = np.array([[f(x) * f(nx) / f(N - 1), f(y) * f(ny) / f(N - 1)] for y in range(N) for x in range(N)], dtype=np.float64)
    p = _f32(undistort_points(pts, K, D, R, P)).reshape(N, N, 2)
    ix0, ix1, iy0, iy1 = p[:, 0, 0].max(), p[:, N - 1, 0].min(), p[0, :, 1].max(), p[N - 1, :, 1].min()
    ox0, ox1, oy0, oy1 = p[..., 0].min(), p[..., 0].max(), p[..., 1].min(), p[..., 1].max()
    sub = lambda a, b: float(f(a) - f(b))
    return (float(ix0), float(iy0), sub(ix1, ix0), sub(iy1, iy0)), (float(ox0), float(oy0), sub(ox1, ox0), sub(oy1, oy0))


def stereo_rectify(K1, D1, K2, D2, image_size, R, T, alpha: float = -1.0, new_image_size=None, zero_disparity: bool = True,
                   return_rois: bool = False):
    """cv::stereoRectify as OpenCV >= 3.4.7 / 4.1.1 computes it.  Returns R1, R2, P1, P2, Q (+ validPixROI1, validPixROI2 as
    (x, y, w, h) with return_rois).  image_size = (width, height) as the reference passes it -- the RECTIFIED frame's size,
    alpha = -1, CALIB_ZERO_DISPARITY by default (python/cam_proj_calibration.py:203-217).

    PINNED against real OpenCV output: called the way the calibration tool behind the reference's data/ESL_calib_hhi.yaml called
    it (camera first, imageSize = (480, 640), newImageSize = (1920, 1080), alpha = 0.5, T in cm) this reproduces the P1, P2, Q,
    validPixROI1 / 2 stored in that file (:90-134) to the last digit (tests/test_calibration_cpu.py).  That includes what the
    alpha = -1 path shares: the common focal length = mean of the two focal lengths x newImageSize / imageSize (older OpenCV took
    the smaller one, shrunk for barrel distortion: the file rules that out), the principal points from the float32 images of the
    four frame corners, five fixed-point rounds in undistortPoints."""
    K1, K2 = np.asarray(K1, np.float64), np.asarray(K2, np.float64)
    R, T = np.asarray(R, np.float64), np.asarray(T, np.float64).reshape(3)
    nx, ny = image_size
    nnx, nny = new_image_size if new_image_size is not None and new_image_size[0] * new_image_size[1] != 0 else image_size
    om = rodrigues(R)
    r_r = rodrigues(-0.5 * om)           # each camera takes half of the relative rotation
    t = r_r @ T
    idx = 0 if abs(t[0]) > abs(t[1]) else 1   # horizontal or vertical stereo
    c, nt = t[idx], np.linalg.norm(t)
    uu = np.zeros(3)
    uu[idx] = 1.0 if c > 0 else -1.0
    ww = np.cross(t, uu)                 # rotate the baseline onto the image axis
    nw = np.linalg.norm(ww)
    if nw > 0:
        ww *= np.arccos(abs(c) / nt) / nw
    wR = rodrigues(ww)
    R1 = wR @ r_r.T
    R2 = wR @ r_r
    t = R2 @ T

    # common focal length: the mean of the two, scaled to the new image size
    ratio = (nnx / nx / 2) if idx == 1 else (nny / ny / 2)
    fc_new = (K1[idx ^ 1, idx ^ 1] + K2[idx ^ 1, idx ^ 1]) * ratio
    # principal points: centre the (rectified) images of the four frame corners; the points are float32 in OpenCV
    corners = np.array([[0, 0], [nx - 1, 0], [0, ny - 1], [nx - 1, ny - 1]], dtype=np.float64)
    cc = []
    for K, D, Rk in ((K1, D1, R1), (K2, D2, R2)):
        n = _f32(undistort_points(corners, K, D))
        xyz = Rk @ np.stack((n[:, 0], n[:, 1], np.ones(4)), 0)
        px, py = _f32(fc_new * xyz[0] / xyz[2]), _f32(fc_new * xyz[1] / xyz[2])
        cc.append(np.array([(nx - 1) / 2 - px.mean(), (ny - 1) / 2 - py.mean()]))
    if zero_disparity:                   # CALIB_ZERO_DISPARITY: the same principal point in both views
        cc[0] = cc[1] = (cc[0] + cc[1]) * 0.5
    else:                                # only along the axis perpendicular to the baseline
        m = (cc[0][idx ^ 1] + cc[1][idx ^ 1]) * 0.5
        cc[0][idx ^ 1] = cc[1][idx ^ 1] = m
    P1 = np.array([[fc_new, 0, cc[0][0], 0], [0, fc_new, cc[0][1], 0], [0, 0, 1, 0]], dtype=np.float64)
    P2 = np.array([[fc_new, 0, cc[1][0], 0], [0, fc_new, cc[1][1], 0], [0, 0, 1, 0]], dtype=np.float64)
    P2[idx, 3] = t[idx] * fc_new
    in1, out1 = _inner_outer_rectangles(K1, D1, R1, P1[:, :3], image_size)
    in2, out2 = _inner_outer_rectangles(K2, D2, R2, P2[:, :3], image_size)
    (cx1_0, cy1_0), (cx2_0, cy2_0) = cc[0], cc[1]
    cx1, cy1, cx2, cy2 = nnx * cx1_0 / nx, nny * cy1_0 / ny, nnx * cx2_0 / nx, nny * cy2_0 / ny
    s = 1.0
    if alpha >= 0:  # 0: only valid pixels stay (inner rectangles), 1: every source pixel stays (outer rectangles)
        def scale(r, cx, cy, cx0, cy0, pick):
            return pick(pick(pick(cx / (cx0 - r[0]), cy / (cy0 - r[1])), (nnx - cx) / (r[0] + r[2] - cx0)), (nny - cy) / (r[1] + r[3] - cy0))
        s0 = max(scale(in1, cx1, cy1, cx1_0, cy1_0, max), scale(in2, cx2, cy2, cx2_0, cy2_0, max))
        s1 = min(scale(out1, cx1, cy1, cx1_0, cy1_0, min), scale(out2, cx2, cy2, cx2_0, cy2_0, min))
        s = s0 * (1 - alpha) + s1 * alpha
    fc_new *= s
    P1[0, 0] = P1[1, 1] = P2[0, 0] = P2[1, 1] = fc_new
    P1[0, 2], P1[1, 2], P2[0, 2], P2[1, 2] = cx1, cy1, cx2, cy2
    P2[idx, 3] *= s
    # (cv::stereoRectify builds Q from the principal points: -cx1, -cy1, f, -1/Tx, (cx1 - cx2)/Tx)
    Q = np.array([[1, 0, 0, -cx1], [0, 1, 0, -cy1], [0, 0, 0, fc_new], [0, 0, -1.0 / t[idx], (cx1 - cx2) / t[idx] if idx == 0 else (cy1 - cy2) / t[idx]]],
                 dtype=np.float64)
    if not return_rois:
        return R1, R2, P1, P2, Q

    def roi(r, cx0, cy0, cx, cy):
        x, y = int(np.ceil((r[0] - cx0) * s + cx)), int(np.ceil((r[1] - cy0) * s + cy))
        w, h = int(np.floor(r[2] * s)), int(np.floor(r[3] * s))
        x0, y0, x1, y1 = max(x, 0), max(y, 0), min(x + w, nnx), min(y + h, nny)  # & Rect(0, 0, newImageSize)
        return (x0, y0, x1 - x0, y1 - y0) if x1 > x0 and y1 > y0 else (0, 0, 0, 0)
    return R1, R2, P1, P2, Q, roi(in1, cx1_0, cy1_0, cx1, cy1), roi(in2, cx2_0, cy2_0, cx2, cy2)


def init_undistort_rectify_map(K, D, R, P, size):
    """cv::initUndistortRectifyMap (CV_32FC1): for every pixel of the rectified image, where to sample the source."""
    w, h = size
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    ir = np.linalg.inv(np.asarray(P, np.float64)[:3, :3] @ np.asarray(R, np.float64))
    x = ir[0, 0] * u + ir[0, 1] * v + ir[0, 2]
    y = ir[1, 0] * u + ir[1, 1] * v + ir[1, 2]
    wz = ir[2, 0] * u + ir[2, 1] * v + ir[2, 2]
    x, y = x / wz, y / wz
    xd, yd = distort_normalized(x, y, D)
    return (K[0, 0] * xd + K[0, 2]).astype(np.float32), (K[1, 1] * yd + K[1, 2]).astype(np.float32)


def init_undistort_rectify_map_inverse(K, D, R, P, size):
    """python/cam_proj_calibration.py:31-41: source pixel -> rectified coordinates, via undistortPoints (f32 in)."""
    w, h = size
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    pts = np.stack((u.ravel(), v.ravel()), -1).astype(np.float32).astype(np.float64)
    out = undistort_points(pts, K, D, R, P).astype(np.float32)
    return out[:, 0].reshape(h, w), out[:, 1].reshape(h, w)


def mapf_to_i16(m: np.ndarray) -> np.ndarray:
    """python/cam_proj_calibration.py:44-48"""
    assert m.dtype == np.float32
    r = np.rint(m)
    assert r.min() >= np.iinfo(np.int16).min and r.max() <= np.iinfo(np.int16).max
    return r.astype(np.int16)


def remap_nearest(img, mapx, mapy, border: str, border_value=0.0):
    """cv::remap(INTER_NEAREST) with BORDER_REPLICATE or BORDER_CONSTANT (python/proj_time_map.py:22-29)."""
    h, w = img.shape
    ix = np.rint(mapx).astype(np.int64)
    iy = np.rint(mapy).astype(np.int64)
    if border == "replicate":
        return img[np.clip(iy, 0, h - 1), np.clip(ix, 0, w - 1)]
    ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
    out = np.full(mapx.shape, border_value, dtype=img.dtype)
    out[ok] = img[iy[ok], ix[ok]]
    return out


# ---- the reference's parameter / maps objects --------------------------------------------------------------
@dataclass
class CamProjCalibrationParams:
    camera_width: int
    camera_height: int
    projector_width: int
    projector_height: int
    rect_image_width: int
    rect_image_height: int
    camera_K: np.ndarray
    camera_D: np.ndarray
    projector_K: np.ndarray
    projector_D: np.ndarray
    cam2proj_R: np.ndarray
    cam2proj_T: np.ndarray

    @staticmethod
    def from_yaml(path, camera_width, camera_height, projector_width, projector_height, rectification_scale=2.75):
        """python/cam_proj_calibration.py:77-108 (projector distortion zeroed, rect = round(2.75 * camera))."""
        data = open_calibration_data(path)
        return CamProjCalibrationParams(
            camera_width, camera_height, projector_width, projector_height,
            round(camera_width * rectification_scale), round(camera_height * rectification_scale),
            read_cv_matrix(data, "camera_intrinsic_matrix"), read_cv_matrix(data, "camera_distortion_coefficients"),
            read_cv_matrix(data, "projector_intrinsic_matrix"), np.zeros((5,)),
            read_cv_matrix(data, "relative_rotation"), read_cv_matrix(data, "relative_translation"))


    @staticmethod
    def from_ESL_yaml(path, camera_width, camera_height, projector_width, projector_height, rectification_scale=3):
        """python/cam_proj_calibration.py:110-140: the ESL dataset's OpenCV-FileStorage calibration (cam_K, cam_kc, proj_K,
        proj_kc, R, T); the rectified frame is 3 x the PROJECTOR size; the projector's distortion is kept."""
        fs = read_opencv_filestorage(path)
        need = ("cam_K", "cam_kc", "proj_K", "proj_kc", "R", "T")
        missing = [k for k in need if k not in fs]
        if missing:
            raise ValueError(f"{path}: missing node(s) {missing}")
        return CamProjCalibrationParams(
            camera_width, camera_height, projector_width, projector_height,
            round(projector_width * rectification_scale), round(projector_height * rectification_scale),
            np.asarray(fs["cam_K"], np.float64), np.asarray(fs["cam_kc"], np.float64),
            np.asarray(fs["proj_K"], np.float64), np.asarray(fs["proj_kc"], np.float64),
            np.asarray(fs["R"], np.float64), np.asarray(fs["T"], np.float64))


def build_eval_tables(cp: CamProjCalibrationParams, **kw) -> dict:
    """The table configuration of the offline evaluation (python/eval/compute_depth_x_maps.py:57-77): CamProjMaps(calib,
    zero_undistort_proj_map=True) and ProjectorTimeMap.from_calib(scan_upwards=False, remap_border_mode=BORDER_CONSTANT)."""
    return build_tables(cp, scan_upwards=False, zero_undistort_proj_map=True, time_map_border="constant", **kw)


def build_tables(cp: CamProjCalibrationParams, z_near=0.1, z_far=1.2, scan_upwards=True, device: int = 0,
                 x_map_on_gpu: bool = True, projector_time_map_rectified: Optional[np.ndarray] = None,
                 zero_undistort_proj_map: bool = False, time_map_border: str = "replicate", x_map_fn=None) -> dict:
    """Everything DepthReprojectionPipe.__post_init__ builds (python/depth_reprojection_pipe.py:69-99), as the
    tables dict XMapsEngine / RuntimeParams.tables take.  Projector = camera 1 of the stereo pair (calib:194-217)."""
    size = (cp.rect_image_width, cp.rect_image_height)
    R1, R2, P1, P2, Q = stereo_rectify(cp.projector_K, cp.projector_D, cp.camera_K, cp.camera_D, size, cp.cam2proj_R,
                                       cp.cam2proj_T)
    # forward maps (rectified pixel -> source pixel) for the projector: used to rectify the time map
    # "ESL compatibility: projector distortion is ignored here, but still used in cv2.stereoRectify" (calib:229-230)
    pmx, pmy = init_undistort_rectify_map(cp.projector_K, np.zeros(5) if zero_undistort_proj_map else cp.projector_D, R2, P2, size)
    # inverse maps (source pixel -> rectified coords), rounded to int16: the per-event LUT and the projector map
    cmx, cmy = init_undistort_rectify_map_inverse(cp.camera_K, cp.camera_D, R1, P1, (cp.camera_width, cp.camera_height))
    qmx, qmy = init_undistort_rectify_map_inverse(cp.projector_K, cp.projector_D, R2, P2,
                                                  (cp.projector_width, cp.projector_height))
    if projector_time_map_rectified is not None:  # ProjectorTimeMap.from_file (python/proj_time_map.py:46-49)
        time_map_rect = np.ascontiguousarray(projector_time_map_rectified, dtype=np.float32)
    else:                                         # ProjectorTimeMap.from_calib (:36-44)
        time_map = generate_linear_projector_time_map(cp.projector_width, cp.projector_height, scan_upwards)
        time_map_rect = remap_nearest(time_map, pmx, pmy, time_map_border)  # BORDER_REPLICATE live, BORDER_CONSTANT in eval
    x_off, xw = 4242, cp.projector_width
    if x_map_fn is not None:  # a caller-supplied builder with the reference's signature (the CPU tests pass the oracle's)
        x_map, _ = x_map_fn(time_map_rect, xw, xw - 1, x_off, cp.projector_width)
    elif x_map_on_gpu:
        from .x_map import compute_x_map_from_time_map
        x_map, _ = compute_x_map_from_time_map(time_map_rect, xw, xw - 1, x_off, cp.projector_width, device=device)
    else:
        raise RuntimeError("the X-map builder runs on the GPU (x_maps_amd.x_map); there is no CPU path in the product")
    return {
        "cam_w": cp.camera_width, "cam_h": cp.camera_height, "proj_w": cp.projector_width, "proj_h": cp.projector_height,
        "rect_w": cp.rect_image_width, "rect_h": cp.rect_image_height,
        "cam_mapx_i16": mapf_to_i16(cmx), "cam_mapy_i16": mapf_to_i16(cmy),
        "proj_x_map": x_map, "disp_proj_mapxy_i16": np.ascontiguousarray(np.stack((mapf_to_i16(qmx), mapf_to_i16(qmy)), -1)),
        "x_map_width": xw, "t_px_scale": xw - 1, "x_offset": x_off, "p03": float(P2[0, 3]),
        "z_near": z_near, "z_far": z_far,
        # kept for tests / rigs
        "R1": R1, "R2": R2, "P1": P1, "P2": P2, "Q": Q, "time_map_rect": time_map_rect,
        # float rectify maps of the camera: rectify_cam_coords_f32 / the point cloud of the evaluation caller (calib:238-245)
        "cam_mapx_f32": cmx, "cam_mapy_f32": cmy,
    }
