"""Deterministic synthetic rig: tables + event frames for the BASELINE.json configs.

Shapes follow SURVEY.md section 8(d):
  C-1M  : camera 640x480, projector 640x480, rectified frame 1760x1320 (= round(2.75 * cam),
          python/cam_proj_calibration.py:84,97-98), X-map 1320x640 (rows = rect height, columns =
          projector width, python/x_maps_disparity.py:58-59), N = 1 000 000 events / frame.
  C-10M : camera = projector 1280x720, rect 3520x1980, X-map 1980x1280, N = 10 000 000.

The tables are an affine stand-in for what cv2.stereoRectify / undistortPoints would produce (OpenCV
is not available offline); they keep every property the hot path depends on: int16 LUT values that
can leave the rectified frame, an X-map with undefined (0) cells incl. column 0
(python/x_map.py:33-34 never fills t == 0), a projector->rect map that partly falls outside the frame.

Events: time-sorted int64 microsecond stamps over a ~13 ms scan with many ties, camera x correlated
with time (the projector scans x-slow), y uniform, ~3 events/pixel => heavy duplicates like real data.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# Metavision EventCD record: 16 bytes, x:u2@0 y:u2@2 p:i2@4 t:i8@8
EVENT_CD_DTYPE = np.dtype(
    {"names": ["x", "y", "p", "t"], "formats": ["<u2", "<u2", "<i2", "<i8"], "offsets": [0, 2, 4, 8], "itemsize": 16}
)

X_OFFSET = 4242


@dataclass
class RigConfig:
    name: str
    cam_w: int
    cam_h: int
    proj_w: int
    proj_h: int
    n_events: int
    rectification_scale: float = 2.75

    @property
    def rect_w(self) -> int:
        return round(self.cam_w * self.rectification_scale)

    @property
    def rect_h(self) -> int:
        return round(self.cam_h * self.rectification_scale)


C_1M = RigConfig("C-1M", 640, 480, 640, 480, 1_000_000)
C_10M = RigConfig("C-10M", 1280, 720, 1280, 720, 10_000_000)
C_TINY = RigConfig("C-tiny", 64, 48, 64, 48, 4_000)


def make_tables(cfg: RigConfig, z_near: float = 0.1, z_far: float = 1.2) -> dict:
    """Affine synthetic tables, all in the reference's dtypes/layouts (row-major int16)."""
    cw, ch, pw, ph = cfg.cam_w, cfg.cam_h, cfg.proj_w, cfg.proj_h
    rw, rh = cfg.rect_w, cfg.rect_h
    sx = rw / cw  # 2.75
    ys, xs = np.mgrid[0:ch, 0:cw].astype(np.float64)
    # camera pixel -> rectified coords (slight shear; a few pixels fall outside the rectified frame)
    cam_mapx = np.rint(2.0 * xs * (sx / 2.75) + 100.0 * (cw / 640) + 0.05 * ys).astype(np.int16)
    # rows: the first/last few camera rows land above / below the rectified frame (y-inlier mask)
    cam_mapy = np.rint(2.8 * ys * (sx / 2.75) - 20.0 * (ch / 480) + 0.02 * xs).astype(np.int16)
    # X-map: rect row, time column -> rect x + X_OFFSET
    yr, tc = np.mgrid[0:rh, 0:pw].astype(np.float64)
    xmap = np.rint(X_OFFSET + 300.0 * (cw / 640) + tc * (rw - 400.0 * (cw / 640)) / pw + 0.02 * yr).astype(np.int16)
    xmap[:, 0] = 0  # t == 0 is never defined by the reference's builder
    xmap[:6, :] = 0  # undefined bands at the top / bottom of the rectified projector image
    xmap[rh - 5:, :] = 0
    xmap[(yr.astype(np.int64) * 131 + tc.astype(np.int64) * 71) % 257 == 0] = 0  # scattered holes
    # projector pixel -> rect coords (x, y interleaved, like cv2's CV_16SC2 map)
    vs, us = np.mgrid[0:ph, 0:pw].astype(np.float64)
    # slightly larger than the rectified frame so the border pixels exercise BORDER_CONSTANT
    pmx = np.rint(us * (rw / pw) * 1.02 - 12.0 * (cw / 640) + 0.03 * vs).astype(np.int16)
    pmy = np.rint(vs * (rh / ph) * 1.01 - 6.0 * (ch / 480) + 0.02 * us).astype(np.int16)
    pmap = np.ascontiguousarray(np.stack((pmx, pmy), axis=-1))
    return {
        "cam_w": cw, "cam_h": ch, "proj_w": pw, "proj_h": ph, "rect_w": rw, "rect_h": rh,
        "cam_mapx_i16": np.ascontiguousarray(cam_mapx),
        "cam_mapy_i16": np.ascontiguousarray(cam_mapy),
        "proj_x_map": np.ascontiguousarray(xmap),
        "disp_proj_mapxy_i16": pmap,
        "x_map_width": pw, "t_px_scale": pw - 1, "x_offset": X_OFFSET,
        # P2[0,3] ~ f_rect * baseline  (f ~ 541 px * 2.75, baseline ~ 0.13 m; data/ESL_calib_hhi.yaml scale)
        "p03": 541.0 * cfg.rectification_scale * 0.13 * (cw / 640),
        "z_near": z_near, "z_far": z_far,
    }


C_SHARED = RigConfig("C-shared", 160, 120, 270, 120, 40_000)


def make_tables_shared_cells(cfg: RigConfig = C_SHARED, cols_per_cell: float = 3.3, slant: float = -0.4, z_near: float = 0.1,
                             z_far: float = 1.2) -> dict:
    """A rig shaped like the reference's own calibration (data/ESL_calib_hhi.yaml through cam_proj_calibration.py:299-303):
    `cols_per_cell` consecutive X-map time columns of a row land on ONE cell of the rectified frame (X_MAP_WIDTH =
    projector_width is finer than the projector's image in the rectified frame), and a time column's cell moves `slant`
    columns per row.  The (row, time column) -> cell map is not injective: such rigs take the owner tiles
    (csrc/xmaps_k1own.hpp).  The camera LUT is made consistent with it: an event at camera x ~ t / scan * cam_w has a
    disparity around 30."""
    tb = make_tables(cfg, z_near, z_far)
    cw, ch, pw = cfg.cam_w, cfg.cam_h, cfg.proj_w
    rw, rh = cfg.rect_w, cfg.rect_h
    ys, xs = np.mgrid[0:ch, 0:cw].astype(np.float64)
    cam_mapy = tb["cam_mapy_i16"].astype(np.float64)
    k = 1.0 / cols_per_cell
    x0 = 18.0 + max(0.0, -slant) * rh
    yr, tc = np.mgrid[0:rh, 0:pw].astype(np.float64)
    xmap = np.rint(X_OFFSET + x0 + k * tc + slant * yr).astype(np.int16)
    xmap[:, 0] = 0
    xmap[:6, :] = 0
    xmap[rh - 5:, :] = 0
    xmap[(yr.astype(np.int64) * 131 + tc.astype(np.int64) * 71) % 257 == 0] = 0
    assert xmap.max() - X_OFFSET < rw
    cam_mapx = np.rint(x0 - 30.0 + k * pw * (xs / cw) + slant * cam_mapy).astype(np.int16)
    tb["proj_x_map"] = np.ascontiguousarray(xmap)
    tb["cam_mapx_i16"] = np.ascontiguousarray(cam_mapx)
    tb["p03"] = 60.0
    return tb


def make_events(cfg: RigConfig, frame: int = 0, n: int | None = None, *, shuffled: bool = False,
                p_zero_fraction: float = 0.0, t0: int = 5_000_000, scan_us: int = 13_000) -> np.ndarray:
    """One frame of EventCD records (structured AoS array, like Metavision hands them over).

    rng = default_rng(20230 + frame); t = t0 + sort(U[0, scan_us)); x ~ t-correlated + N(0, 2 px);
    y ~ U[0, cam_h).  `shuffled=True` is the adversarial raster/unsorted variant (x, y i.i.d. uniform,
    t permuted).  `p_zero_fraction` flips that share of polarities to 0 to exercise the polarity mask.
    """
    n = cfg.n_events if n is None else n
    rng = np.random.default_rng(20230 + frame)
    evs = np.zeros(n, dtype=EVENT_CD_DTYPE)
    if n == 0:
        return evs
    t_rel = np.sort(rng.integers(0, scan_us, n))
    if shuffled:
        x = rng.integers(0, cfg.cam_w, n)
        t_rel = rng.permutation(t_rel)
    else:
        x = np.clip(np.rint(t_rel / scan_us * cfg.cam_w + rng.normal(0.0, 2.0, n)), 0, cfg.cam_w - 1)
    evs["x"] = x.astype(np.uint16)
    evs["y"] = rng.integers(0, cfg.cam_h, n).astype(np.uint16)
    evs["t"] = t0 + t_rel
    evs["p"] = 1
    if p_zero_fraction > 0:
        evs["p"][rng.random(n) < p_zero_fraction] = 0
    return evs


def to_soa(evs: np.ndarray):
    """AoS EventCD -> contiguous SoA columns (x u16, y u16, t i64, p i16)."""
    return (np.ascontiguousarray(evs["x"]), np.ascontiguousarray(evs["y"]),
            np.ascontiguousarray(evs["t"]), np.ascontiguousarray(evs["p"]))
