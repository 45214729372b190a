"""-m gpu: randomized differential test of the fused path against the NumPy oracle: random table shapes and contents
(LUT entries outside the rectified frame, undefined X-map cells, maps pointing outside, odd sizes), random event streams
(sorted / unsorted, ties, duplicates, with / without a polarity column, SoA / AoS), both views, with and without the
time-sorted declaration.  Sizes are drawn so that both K1 variants (LDS-tiled and one-thread-per-event) are exercised."""
import numpy as np
import pytest

import xmaps_oracle as O
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    cam_w, cam_h = int(rng.integers(5, 97)), int(rng.integers(4, 73))
    rect_w, rect_h = int(rng.integers(cam_w, 3 * cam_w + 8)), int(rng.integers(cam_h + 2, 3 * cam_h + 8))
    proj_w, proj_h = int(rng.integers(3, 90)), int(rng.integers(3, 70))
    xmap_w = int(rng.integers(2, 120))
    ys, xs = np.mgrid[0:cam_h, 0:cam_w]
    sx, sy = rect_w / cam_w, rect_h / cam_h
    mapx = np.rint(rng.uniform(0.5, 1.0) * sx * xs + rng.uniform(-6, 6) + rng.uniform(-0.1, 0.1) * ys).astype(np.int16)
    mapy = np.rint(rng.uniform(0.7, 1.1) * sy * ys + rng.uniform(-5, 3) + rng.uniform(-0.1, 0.1) * xs).astype(np.int16)
    xmap_h = max(3, rect_h + int(rng.integers(-3, 4)))  # the X-map may have more / fewer rows than the rectified frame
    yr, tc = np.mgrid[0:xmap_h, 0:xmap_w]
    xmap = np.rint(4242 + rng.uniform(0, 0.3) * rect_w + tc * rng.uniform(0.3, 0.9) * rect_w / xmap_w
                   + rng.uniform(-0.15, 0.15) * yr).astype(np.int16)
    xmap[rng.random(xmap.shape) < rng.uniform(0, 0.2)] = 0
    if rng.random() < 0.5:
        xmap[:, 0] = 0
    vs, us = np.mgrid[0:proj_h, 0:proj_w]
    pm = np.stack((np.rint(us * rect_w / proj_w * rng.uniform(0.8, 1.2) + rng.uniform(-5, 5)),
                   np.rint(vs * rect_h / proj_h * rng.uniform(0.8, 1.2) + rng.uniform(-5, 5))), -1).astype(np.int16)
    tb = {"cam_w": cam_w, "cam_h": cam_h, "proj_w": proj_w, "proj_h": proj_h, "rect_w": rect_w, "rect_h": rect_h,
          "cam_mapx_i16": mapx, "cam_mapy_i16": mapy, "proj_x_map": np.ascontiguousarray(xmap),
          "disp_proj_mapxy_i16": np.ascontiguousarray(pm), "t_px_scale": xmap_w - 1, "x_offset": 4242,
          "p03": float(rng.uniform(5, 300)) * (1 if rng.random() < 0.9 else -1), "z_near": 0.1, "z_far": float(rng.uniform(0.5, 3.0))}
    # event count: sometimes enough for the tiled kernel (>= 1024 * xmap_w / 3.5), sometimes tiny
    dense = rng.random() < 0.6
    n = int(rng.integers(300 * xmap_w, 600 * xmap_w)) if dense else int(rng.integers(1, 3000))
    n = min(n, 80_000)
    span = int(rng.integers(1, 20_000))
    t_rel = np.sort(rng.integers(0, span, n))
    if rng.random() < 0.3:
        t_rel = rng.permutation(t_rel)  # unsorted (raster-order filters)
    if rng.random() < 0.7:
        x = np.clip(np.rint(t_rel / span * cam_w + rng.normal(0, rng.uniform(0.5, 4), n)), 0, cam_w - 1)
    else:
        x = rng.integers(0, cam_w, n)
    evs = np.zeros(n, S.EVENT_CD_DTYPE)
    evs["x"], evs["y"] = x, rng.integers(0, cam_h, n)
    evs["t"] = int(rng.integers(0, 2 ** 40)) + t_rel
    evs["p"] = (rng.random(n) < 0.93) if rng.random() < 0.5 else 1
    return tb, evs, bool(rng.random() < 0.5), bool(rng.random() < 0.5)


@pytest.mark.parametrize("seed", range(120))
def test_random_tables_and_streams(seed):
    tb, evs, camera, declare_sorted = _random_case(seed)
    x, y, t, p = S.to_soa(evs)
    use_p = bool((p != 1).any())
    keep = p == 1
    try:
        ref = O.process_ev_frame(tb, x[keep].astype(np.int64), y[keep].astype(np.int64), t[keep], camera_perspective=camera)
        ref_err = None
    except (IndexError, ValueError) as e:  # the reference would raise: the build must report the same class of error
        ref, ref_err = None, type(e)
    with XMapsEngine(tb, camera_perspective=camera, assume_time_sorted=declare_sorted, n_slots=2) as eng:
        if ref_err is IndexError:
            with pytest.raises(IndexError):
                eng.process_frame(x, y, t, p if use_p else None)
            return
        if ref_err is ValueError:  # every event filtered out by polarity: t.min() of an empty array
            d, b, st = eng.process_frame(x, y, t, p if use_p else None)
            assert st.n_used == 0 and not d.any()
            return
        d, b, st = eng.process_frame(x, y, t, p if use_p else None)
        assert st.n_used == int(keep.sum()) and st.n_inliers == int(ref["mask"].sum())
        assert np.array_equal(d, ref["depth"]) and np.array_equal(b, ref["bgr"])
        d2, b2, st2 = eng.process_events(evs, use_polarity=use_p)
        assert np.array_equal(d2, ref["depth"]) and np.array_equal(b2, ref["bgr"]) and st2.n_inliers == st.n_inliers
