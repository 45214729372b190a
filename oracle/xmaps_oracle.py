"""CPU oracle (NumPy) for the X-maps per-event disparity-lookup / depth-reprojection hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker / the timed CPU baseline.
The product path (`x_maps_amd`) never imports anything from `oracle/`.

It is a restatement -- in this repo's own words -- of the reference's algorithm, function by
function, keeping the reference's *pass structure* (one NumPy pass per reference statement) so that
it doubles as the like-for-like "NumPy, 1 core" CPU baseline.  Each function cites the reference
file:line it follows (paths relative to /root/reference).

Parity status (see DESIGN.md "Oracle pinning"):
  * A1, A2, A3, A3', A5, A6, white-mask, X-map construction, linear time map: PINNED against golden
    vectors produced here by importing the reference's own functions (tests/golden/make_golden.py).
  * A4 (cv2.dilate + cv2.remap) and A7 (cv2.applyColorMap TURBO): third-party OpenCV arithmetic, OpenCV
    is not available offline and the reference has no tests -> "parity unpinned"; semantics restated
    from the OpenCV documentation (max filter that ignores the border; nearest gather with constant-0
    border; Turbo table from Google's published floats).
"""
from __future__ import annotations

import numpy as np

X_OFFSET = 4242  # python/x_maps_disparity.py:49

# --------------------------------------------------------------------------------------------------
# packed "last-writer-wins" key used by the GPU path and by the sharded (multi-GPU) reduce.
#   bits  0..15  disparity (int16 >= 0 reinterpreted as u16)
#   bits 16..43  event index inside the frame (28 bits)
#   bits 44..62  frame tag (19 bits, >= 1; 0 = cell never written).  bit 63 stays 0 so that a signed
#                int64 MAX all-reduce orders keys exactly like the unsigned compare on the GPU.
KEY_DISP_BITS = 16
KEY_IDX_BITS = 28
KEY_TAG_BITS = 19
KEY_IDX_SHIFT = KEY_DISP_BITS
KEY_TAG_SHIFT = KEY_DISP_BITS + KEY_IDX_BITS
KEY_MAX_EVENTS = 1 << KEY_IDX_BITS
KEY_MAX_TAG = (1 << KEY_TAG_BITS) - 1


# --------------------------------------------------------------------------------------------------
# A1  rectification LUT gather
def rectify_cam_coords_i16(mapx_i16, mapy_i16, x, y):
    """(x, y) camera pixel -> rounded rectified coordinates, int16.

    Follows CamProjMaps.rectify_cam_coords_i16, python/cam_proj_calibration.py:277-281:
    two fancy-index gathers `map[y, x]`.  NumPy index rules apply (IndexError when out of range).
    """
    xr = mapx_i16[y, x]
    yr = mapy_i16[y, x]
    return xr, yr


# --------------------------------------------------------------------------------------------------
# A2  time normalisation + X-map gather + disparity + inlier masks
def time_to_xmap_column(t, t_px_scale):
    """t -> X-map column, int16.  python/x_maps_disparity.py:12-19.

    int64 `t` : (t - tmin) is int64, `/` promotes both sides to float64 (true division), `* S`
    stays float64, np.rint is round-half-even, the int16 cast truncates an already-integral value.
    float32/float64 `t` (eval caller, python/eval/compute_depth_x_maps.py:89-96) stay in their dtype.
    tmax == tmin gives 0/0 = NaN; NumPy's NaN -> int16 cast yields 0 on x86-64 (observed here); the
    build defines that as the behaviour: every event lands in column 0.
    """
    t = np.asarray(t)
    tmin = t.min()  # ValueError on an empty frame, as in the reference (xmd:12)
    tmax = t.max()
    with np.errstate(invalid="ignore", divide="ignore"):
        tn = (t - tmin) / (tmax - tmin)
        scaled = np.rint(tn * t_px_scale)
    if tmax == tmin:
        return np.zeros(t.shape, dtype=np.int16)
    return scaled.astype(np.int16)


def compute_disparity(xr_i16, yr_i16, t, proj_x_map, t_px_scale, x_offset=X_OFFSET):
    """Per-event disparity by X-map lookup.  python/x_maps_disparity.py:9-32.

    Returns (disp[M] int16 compacted, inlier_mask[N] bool).
      m_y  = 0 <= yr < H-1                      (xmd:23 -- the last X-map row is excluded)
      xp   = X[yr, ts]                          (xmd:25)
      disp = xp - xr - x_offset  in int16 wrap-around arithmetic (xmd:27)
      m_d  = disp >= 0                          (xmd:29)
    """
    ts = time_to_xmap_column(t, t_px_scale)
    m = (yr_i16 >= 0) & (yr_i16 < proj_x_map.shape[0] - 1)
    xp = proj_x_map[yr_i16[m], ts[m]]
    # int16 - int16 - (python int) stays int16 and wraps (NumPy 1.x value-based casting and NumPy 2
    # NEP-50 agree here because 4242 fits int16)
    with np.errstate(over="ignore"):
        disp = (xp - xr_i16[m] - np.int16(x_offset)).astype(np.int16)
    md = disp >= 0
    mask = m.copy()
    mask[m] = md
    return disp[md], mask


# --------------------------------------------------------------------------------------------------
# A3 / A3'  disparity-frame scatter, last writer wins
def disp_map_projector_view(xr_i16, yr_i16, inlier_mask, disp, rect_h, rect_w):
    """Rectified disparity frame seen from the projector.  python/cam_proj_calibration.py:299-303.

    column = int16(xr + disp) (= xp - x_offset), row = yr; NumPy fancy assignment: negative indices
    wrap once, anything else out of range raises IndexError, duplicates -> the LAST event wins.
    """
    with np.errstate(over="ignore"):
        xpr = np.rint(xr_i16[inlier_mask] + disp).astype(np.int16)
    frame = np.zeros((rect_h, rect_w), dtype=np.float32)
    frame[yr_i16[inlier_mask], xpr] = disp
    return frame


def disp_map_camera_view(x, y, inlier_mask, disp, cam_h, cam_w):
    """Disparity frame at the camera pixels.  python/cam_proj_calibration.py:312-317."""
    frame = np.zeros((cam_h, cam_w), dtype=np.float32)
    frame[y[inlier_mask], x[inlier_mask]] = disp
    return frame


# --------------------------------------------------------------------------------------------------
# A4  7x7 dilate + nearest remap (OpenCV semantics restated; parity unpinned)
def dilate7x7(frame):
    """cv2.dilate(frame, ones((7,7))) -- python/disp_to_depth.py:74,86.

    Max over the 7x7 window centred on the pixel; window cells outside the image are ignored
    (OpenCV's default morphology border).  Separable: 7-tap row max then 7-tap column max.
    """
    h, w = frame.shape
    neg = np.float32(-np.inf)
    pad = np.full((h, w + 6), neg, dtype=np.float32)
    pad[:, 3:3 + w] = frame
    rows = pad[:, 0:w].copy()
    for k in range(1, 7):
        np.maximum(rows, pad[:, k:k + w], out=rows)
    pad2 = np.full((h + 6, w), neg, dtype=np.float32)
    pad2[3:3 + h, :] = rows
    out = pad2[0:h, :].copy()
    for k in range(1, 7):
        np.maximum(out, pad2[k:k + h, :], out=out)
    return out


def remap_nearest_i16(frame, mapxy_i16):
    """cv2.remap(frame, map1=mapxy_i16 (H,W,2: x then y), INTER_NEAREST, BORDER_CONSTANT 0).

    python/disp_to_depth.py:89-95.  out[v,u] = frame[my, mx] when 0<=mx<W_r and 0<=my<H_r else 0.
    """
    h, w = frame.shape
    mx = mapxy_i16[..., 0].astype(np.int32)
    my = mapxy_i16[..., 1].astype(np.int32)
    ok = (mx >= 0) & (mx < w) & (my >= 0) & (my < h)
    out = np.zeros(mapxy_i16.shape[:2], dtype=frame.dtype)
    out[ok] = frame[my[ok], mx[ok]]
    return out


def remap_rectified_disp_map_to_proj(rect_disp, disp_proj_mapxy_i16):
    """DisparityToDepth.remap_rectified_disp_map_to_proj, python/disp_to_depth.py:76-97."""
    return remap_nearest_i16(dilate7x7(rect_disp), disp_proj_mapxy_i16)


# --------------------------------------------------------------------------------------------------
# A5  disparity -> depth
def disparity_to_depth_rectified(disp_frame, p03):
    """depth = 0 where disp == 0 else max(P[0,3] / disp, 1e-9), FP64 divide stored as f32.

    python/disp_to_depth.py:46-63 (Numba body; P is float64 so the divide is double precision).
    """
    d = disp_frame.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.maximum(np.float64(p03) / d, 1e-9)
    return np.where(disp_frame == 0, np.float32(0), q.astype(np.float32)).astype(np.float32)


# --------------------------------------------------------------------------------------------------
# A6  clip + normalise to u8
def clip_normalize_uint8_depth_frame(depth, z_near, z_far, mul_in_f64=True):
    """python/disp_to_depth.py:7-21.

    v != 0: clamp to [z_near, z_far] in f32, (v - lo) / (hi - lo) in f32, then `* 255` -- the literal
    255 is an int64 in Numba, so the product is formed in float64 (Numba unifies f32 x i64 -> f64),
    then truncated to u8.  Run as plain Python under NumPy 2 (the stub import used to make the golden
    vectors) the product stays f32; `mul_in_f64=False` reproduces that variant.  The two differ only
    when f32 rounding lifts q*255 onto an integer (<= 1 LSB of the 8-bit visualisation).
    """
    lo = np.float32(z_near)
    hi = np.float32(z_far)
    rng = np.float32(hi - lo)
    v = np.minimum(np.maximum(depth.astype(np.float32), lo), hi)
    q = ((v - lo) / rng).astype(np.float32)
    s = q.astype(np.float64) * 255.0 if mul_in_f64 else (q * np.float32(255)).astype(np.float32)
    out = s.astype(np.uint8)  # truncation, values are in [0, 255]
    out[depth == 0] = 0
    return out


# --------------------------------------------------------------------------------------------------
# A7  colourise (Turbo, BGR) + white mask (parity unpinned vs OpenCV's table rounding)
def _turbo_bgr():
    import importlib.util
    import os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "x_maps_amd", "turbo_lut.py")
    spec = importlib.util.spec_from_file_location("_xm_turbo_lut", p)  # table = data, not product code
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.TURBO_BGR_U8


def generate_color_map(norm_u8):
    """cv2.applyColorMap(u8, COLORMAP_TURBO) then white where u8 == 0.  python/disp_to_depth.py:24-43."""
    bgr = _turbo_bgr()[norm_u8]
    bgr[norm_u8 == 0] = 255
    return bgr


def colorize_depth_from_disp(disp_frame, p03, z_near, z_far):
    """DisparityToDepth.colorize_depth_from_disp, python/disp_to_depth.py:99-115."""
    depth = disparity_to_depth_rectified(disp_frame, p03)
    return generate_color_map(clip_normalize_uint8_depth_frame(depth, z_near, z_far))


# --------------------------------------------------------------------------------------------------
# A8  the whole frame (DepthReprojectionPipe.process_ev_frame, python/depth_reprojection_pipe.py:121-167,
# default NoFilter branch)
def process_ev_frame(tables, x, y, t, camera_perspective=False, want_bgr=True):
    """Returns dict(xr, yr, disp, mask, disp_map, proj_disp (projector view only), depth, bgr)."""
    xr, yr = rectify_cam_coords_i16(tables["cam_mapx_i16"], tables["cam_mapy_i16"], x, y)
    disp, mask = compute_disparity(xr, yr, t, tables["proj_x_map"], tables["t_px_scale"], tables["x_offset"])
    out = {"xr": xr, "yr": yr, "disp": disp, "mask": mask}
    if camera_perspective:
        dm = disp_map_camera_view(x, y, mask, disp, tables["cam_h"], tables["cam_w"])
        out["disp_map"] = dm
        final = dm
    else:
        dm = disp_map_projector_view(xr, yr, mask, disp, tables["rect_h"], tables["rect_w"])
        out["disp_map"] = dm
        final = remap_rectified_disp_map_to_proj(dm, tables["disp_proj_mapxy_i16"])
        out["proj_disp"] = final
    out["depth"] = disparity_to_depth_rectified(final, tables["p03"])
    if want_bgr:
        out["bgr"] = generate_color_map(
            clip_normalize_uint8_depth_frame(out["depth"], tables["z_near"], tables["z_far"]))
    return out


# --------------------------------------------------------------------------------------------------
# packed-key view of A3/A3' (what one GPU shard produces; max-reduce over shards == single-GPU frame)
def key_frame(tables, x, y, t, tmin, tmax, idx_offset=0, tag=1, camera_perspective=False):
    """u64 key per cell = (tag << 44) | (global event index << 16) | disp, max over the events
    that hit the cell; 0 where nothing hit.  `tmin/tmax` are the FRAME's extrema (a shard sees only a
    slice of the events).  Decoding `key & 0xFFFF` where `key >> 44 == tag` gives A3's frame exactly,
    because 'largest event index wins' is NumPy's last-writer-wins.
    """
    xr, yr = rectify_cam_coords_i16(tables["cam_mapx_i16"], tables["cam_mapy_i16"], x, y)
    t = np.asarray(t)
    if tmax == tmin:
        ts = np.zeros(t.shape, np.int16)
    else:
        ts = np.rint(((t - tmin) / (tmax - tmin)) * tables["t_px_scale"]).astype(np.int16)
    X = tables["proj_x_map"]
    m = (yr >= 0) & (yr < X.shape[0] - 1)
    xp = X[yr[m], ts[m]]
    with np.errstate(over="ignore"):
        disp = (xp - xr[m] - np.int16(tables["x_offset"])).astype(np.int16)
    md = disp >= 0
    mask = m.copy()
    mask[m] = md
    disp = disp[md]
    idx = (np.nonzero(mask)[0] + idx_offset).astype(np.uint64)
    keys = (np.uint64(tag) << np.uint64(KEY_TAG_SHIFT)) | (idx << np.uint64(KEY_IDX_SHIFT)) | disp.astype(np.uint64)
    if camera_perspective:
        shape = (tables["cam_h"], tables["cam_w"])
        rows, cols = y[mask].astype(np.int64), x[mask].astype(np.int64)
    else:
        shape = (tables["rect_h"], tables["rect_w"])
        with np.errstate(over="ignore"):
            cols = (xr[mask] + disp).astype(np.int16).astype(np.int64)
        rows = yr[mask].astype(np.int64)
        cols = np.where(cols < 0, cols + shape[1], cols)  # NumPy negative-index wrap
        if ((cols < 0) | (cols >= shape[1])).any():
            raise IndexError("projector-view column out of range")
    kf = np.zeros(shape, dtype=np.uint64)
    np.maximum.at(kf, (rows, cols), keys)
    return kf


def decode_key_frame(kf, tag=1):
    """key frame -> f32 disparity frame (0 where empty / stale tag)."""
    live = (kf >> np.uint64(KEY_TAG_SHIFT)) == np.uint64(tag)
    return np.where(live, (kf & np.uint64(0xFFFF)).astype(np.float32), np.float32(0)).astype(np.float32)


# --------------------------------------------------------------------------------------------------
# setup-time tables ("next" row N1; used here to build realistic X-maps for tests)
def rectify_cam_coords_f32(mapx_f32, mapy_f32, x, y):
    """python/cam_proj_calibration.py:272-275"""
    x = np.asarray(x).astype(np.int64)
    y = np.asarray(y).astype(np.int64)
    return np.asarray(mapx_f32, np.float32)[y, x], np.asarray(mapy_f32, np.float32)[y, x]


def construct_point_cloud(Q, xpr_f32, ypr_f32, disp_f32):
    """python/cam_proj_calibration.py:319-331: homogeneous [x+d, y, -d, 1] through Q (float32), perspective divide,
    y and z negated.  d == 0 divides by zero exactly as there (inf / nan rows)."""
    n = len(xpr_f32)
    pts = np.ones((n, 4), np.float32)
    pts[:, 0] = np.asarray(xpr_f32, np.float32) + np.asarray(disp_f32, np.float32)
    pts[:, 1] = ypr_f32
    pts[:, 2] = -np.asarray(disp_f32, np.float32)
    pc = (np.asarray(Q).astype(np.float32) @ pts.T).T
    with np.errstate(divide="ignore", invalid="ignore"):
        pc = (pc / pc[:, 3:])[:, :3]
    pc[:, 1] = -pc[:, 1]
    pc[:, 2] = -pc[:, 2]
    return pc


def time_surface_to_events(cam_image):
    """python/eval/compute_depth_x_maps.py:83-97: the evaluation caller's event list (raster order, float t in (0, 1])."""
    img = np.array(cam_image, dtype=np.float64, copy=True)
    nz = img != 0
    lo, hi = img[nz].min(), img[nz].max()
    img = (img - lo) / (hi - lo)
    img[img < 0] = 0
    yx = np.argwhere(img > 0)
    return yx[:, 1], yx[:, 0], img[img > 0]


def generate_linear_projector_time_map(proj_w, proj_h, scan_upwards):
    """Ideal raster time map: x is the slow axis, y the fast one.  python/proj_time_map.py:6-19."""
    ys, xs = np.mgrid[0:proj_h, 0:proj_w]
    if scan_upwards:
        ys = ys[::-1]
    return ((xs * proj_h + ys) / (proj_w * proj_h)).astype(np.float32)


def compute_x_map_from_time_map(time_map, x_map_width, t_px_scale, x_offset, num_scanlines):
    """X-map (y, t_col) -> x + x_offset by exhaustive per-row arg-min of |t - time_map[y, x]|.

    python/x_map.py:5-55: t = t_col / t_px_scale (float64); t == 0 is skipped; cells with
    time_map == 0 are skipped; FIRST minimum wins (strict <); kept only if the minimum is
    <= 2 / num_scanlines.  |t - m| is evaluated in float64 with m widened from f32, as Numba does.
    Vectorised per row (broadcast W_t x W_r), same arithmetic and tie rule as the scalar loops.
    """
    h, w = time_map.shape
    x_map = np.zeros((h, x_map_width), dtype=np.int16)
    t_diffs = np.zeros((h, x_map_width), dtype=np.float32)
    tcol = np.arange(x_map_width, dtype=np.float64) / t_px_scale
    max_t_diff = 2 / num_scanlines
    for yy in range(h):
        row = time_map[yy].astype(np.float64)
        valid = row != 0
        if not valid.any():
            continue
        d = np.abs(tcol[:, None] - row[None, :])
        d[:, ~valid] = np.inf
        best = d.argmin(axis=1)  # first minimum
        dmin = d[np.arange(x_map_width), best]
        keep = (tcol != 0) & (dmin <= max_t_diff)
        x_map[yy, keep] = (best[keep] + x_offset).astype(np.int16)
        t_diffs[yy, keep] = dmin[keep].astype(np.float32)
    return x_map, t_diffs


# ---- N4: evaluation metrics (python/eval/create_evaluation_table.py:14-63) ---------------------------------------------------
def load_and_filter(result, gt, min_depth, max_depth):
    result = np.array(result, copy=True)
    result[result >= max_depth] = 0
    result[result <= min_depth] = 0
    result[gt == 0] = 0
    return result


def evaluation_stats(estimate, groundtruth):
    """-> dict(fillrate, rmse, perc_1, perc_5, perc_10, margin), the arithmetic of class evaluation_stats."""
    gt, est = np.asarray(groundtruth), np.asarray(estimate)
    with np.errstate(all="ignore"):
        margin = 0.01 * np.sum(gt[gt > 0]) / np.sum(gt > 0)
        diff = np.abs(gt - est)
        diff[gt == 0] = 0
        fillrate = (np.sum(diff < margin) - np.sum(gt == 0)) / (diff.shape[0] * diff.shape[1] - np.sum(gt == 0))
        diff_sq = pow(gt - est, 2)
        valid = (gt > 0) & (est > 0)
        rmse = np.sqrt(np.sum(diff_sq[valid]) / np.sum(valid)) if np.sum(valid) > 0 else 0
        n = diff.shape[0] * diff.shape[1]
    return {"fillrate": float(fillrate), "rmse": float(rmse), "perc_1": float(100 * np.sum(diff > 1) / n),
            "perc_5": float(100 * np.sum(diff > 5) / n), "perc_10": float(100 * np.sum(diff > 10) / n), "margin": float(margin)}
