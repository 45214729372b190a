#!/usr/bin/env python3
"""Pin-on-arrival for the rows whose arithmetic lives in third-party code that is NOT in this build's container.

The reference calls OpenCV (`cv2.dilate`, `cv2.remap`, `cv2.applyColorMap`, `cv2.stereoRectify`, `cv2.initUndistortRectifyMap`,
`cv2.undistortPoints`: python/disp_to_depth.py:36,86-95, python/cam_proj_calibration.py:211-270, python/proj_time_map.py:22-29)
and Metavision (the RAW reader behind python/bias_events_iterator.py:83-90, `ActivityNoiseFilterAlgorithm` at
python/depth_reprojection_pipe.py:65-67,116-117).  Neither is installed here, so the oracle's restatements of those calls are
"unpinned" (DESIGN.md section 5).  Run this ONCE on a machine that has them -- a reference installation:

    python tools/pin_thirdparty.py            # writes tests/golden/g9_cv2.npz and / or tests/golden/g10_metavision.npz

and commit the files.  tests/test_thirdparty_pins.py skips while they are absent and, once they exist, compares the oracle
(CPU tests) AND the HIP path (-m gpu tests) with them.  The script imports numpy, cv2 and metavision_* only -- never a file of
the reference, never this package (a maintainer's machine need not have a GPU) -- and reads one fixture of this repository
(tests/golden/g6_esl_calib.npz: the matrices of the reference's calibration file).  Everything it feeds the libraries is
generated here from fixed seeds, so the fixtures are reproducible.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
EVENT_CD = np.dtype({"names": ["x", "y", "p", "t"], "formats": ["<u2", "<u2", "<i2", "<i8"], "offsets": [0, 2, 4, 8], "itemsize": 16})


# ---- OpenCV -----------------------------------------------------------------------------------------------------------------
def pin_cv2():
    import cv2
    rng = np.random.default_rng(9)
    out = {"cv2_version": np.array(cv2.__version__)}

    # A4, first half: cv2.dilate(D, ones((7, 7), uint8)) on a sparse float32 frame of integer disparities (d2d:86), values on the
    # border rows / columns included (the border rule is what the restatement had to assume)
    H, W = 96, 132
    d = np.zeros((H, W), np.float32)
    idx = rng.integers(0, H * W, 900)
    d.ravel()[idx] = rng.integers(1, 400, len(idx)).astype(np.float32)
    d[0, :7], d[-1, -7:], d[:5, 0], d[-5:, -1] = 11, 12, 13, 14
    out["dilate_in"] = d
    out["dilate_out"] = cv2.dilate(d, np.ones((7, 7), np.uint8))

    # A4, second half: cv2.remap(frame, map1 = int16 (H, W, 2), map2 = None, INTER_NEAREST, BORDER_CONSTANT) (d2d:89-95): targets
    # inside, on the edge and outside the frame (negative and >= size)
    ph, pw = 80, 70
    m = np.stack((rng.integers(-6, W + 6, (ph, pw)), rng.integers(-6, H + 6, (ph, pw))), axis=-1).astype(np.int16)
    m[0, :4] = [[0, 0], [W - 1, H - 1], [W, 0], [-1, 5]]
    out["remap_src"] = out["dilate_out"]
    out["remap_map_i16"] = m
    out["remap_out"] = cv2.remap(out["dilate_out"], map1=m, map2=None, interpolation=cv2.INTER_NEAREST, borderMode=cv2.BORDER_CONSTANT)

    # proj_time_map.py:22-29: cv2.remap with two float32 maps, INTER_NEAREST, both border modes the reference passes -- the
    # rounding of x.5 targets is what matters
    fx = (rng.random((60, 50)) * (W + 8) - 4).astype(np.float32)
    fy = (rng.random((60, 50)) * (H + 8) - 4).astype(np.float32)
    fx[0, :6] = [0.5, 1.5, 2.5, -0.5, W - 1.5, W - 0.5]
    fy[0, :6] = [0.5, 1.5, 2.5, -0.5, H - 1.5, H - 0.5]
    src = rng.random((H, W)).astype(np.float32)
    out["remapf_src"], out["remapf_mapx"], out["remapf_mapy"] = src, fx, fy
    out["remapf_out_constant"] = cv2.remap(src, fx, fy, cv2.INTER_NEAREST, borderMode=cv2.BORDER_CONSTANT)
    out["remapf_out_replicate"] = cv2.remap(src, fx, fy, cv2.INTER_NEAREST, borderMode=cv2.BORDER_REPLICATE)

    # A7: cv2.applyColorMap(u8, COLORMAP_TURBO) for every input value (d2d:36)
    out["turbo_bgr"] = cv2.applyColorMap(np.arange(256, dtype=np.uint8).reshape(256, 1), cv2.COLORMAP_TURBO).reshape(256, 3)

    # Table builder (cam_proj_calibration.py:174-270) on the reference's own calibration matrices
    g = np.load(os.path.join(GOLDEN, "g6_esl_calib.npz"))
    cam_w, cam_h, proj_w, proj_h = 640, 480, 1080, 1920
    scale = 2.75  # rectification_scale of CamProjCalibrationParams.from_yaml
    rect_w, rect_h = int(cam_w * scale), int(cam_h * scale)
    camK, camD, prjK, prjD = g["camera_K"], g["camera_D"], g["projector_K"], g["projector_D"]
    R1, R2, P1, P2, Q, roi1, roi2 = cv2.stereoRectify(cameraMatrix1=camK, distCoeffs1=camD, cameraMatrix2=prjK, distCoeffs2=prjD,
                                                      imageSize=(rect_w, rect_h), R=g["R"], T=g["T"], alpha=-1)
    out.update(rect_size=np.array([rect_w, rect_h]), R1=R1, R2=R2, P1=P1, P2=P2, Q=Q, roi1=np.array(roi1), roi2=np.array(roi2))

    def inverse_map(K, D, R, P, size):  # initUndistortRectifyMapInverse of the reference, restated (calib:31-41)
        w, h = size
        coords = np.stack(np.meshgrid(np.arange(w), np.arange(h))).reshape((2, -1)).T.reshape((-1, 1, 2)).astype("float32")
        pts = cv2.undistortPoints(coords, K, D, None, R, P)
        maps = pts.reshape((h, w, 2))
        return maps[..., 0], maps[..., 1]

    cmx, cmy = inverse_map(camK, camD, R1, P1, (cam_w, cam_h))          # disp_cam_map{x,y}: A1's LUT before rounding
    out["cam_inv_mapx_f32"], out["cam_inv_mapy_f32"] = cmx, cmy
    out["cam_inv_mapx_i16"], out["cam_inv_mapy_i16"] = np.rint(cmx).astype(np.int16), np.rint(cmy).astype(np.int16)
    pmx, pmy = inverse_map(prjK, prjD, R2, P2, (proj_w, proj_h))        # disp_proj_mapxy: A4's map
    out["proj_inv_mapx_i16_s8"], out["proj_inv_mapy_i16_s8"] = np.rint(pmx).astype(np.int16)[::8, ::8], np.rint(pmy).astype(np.int16)[::8, ::8]
    fmx, fmy = cv2.initUndistortRectifyMap(prjK, np.zeros(5), R2, P2, (rect_w, rect_h), cv2.CV_32FC1)  # projector_map{x,y} (calib:237-244)
    out["proj_fwd_mapx_f32_s16"], out["proj_fwd_mapy_f32_s16"] = fmx[::16, ::16], fmy[::16, ::16]
    path = os.path.join(GOLDEN, "g9_cv2.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB, OpenCV {cv2.__version__})")


# ---- Metavision -------------------------------------------------------------------------------------------------------------
def evt3_words_singles(ev):
    """one event = [TIME_HIGH / TIME_LOW when they change] [ADDR_Y when it changes] ADDR_X (EVT 3.0, public format): the plainest
    legal stream"""
    w = []
    hi = lo = y = None
    for x_, y_, p_, t_ in zip(ev["x"], ev["y"], ev["p"], ev["t"]):
        t_ = int(t_)
        h12, l12 = (t_ >> 12) & 0xFFF, t_ & 0xFFF
        if h12 != hi:
            w.append(0x8000 | h12)
            hi, lo = h12, None
        if l12 != lo:
            w.append(0x6000 | l12)
            lo = l12
        if int(y_) != y:
            w.append(0x0000 | int(y_))
            y = int(y_)
        w.append(0x2000 | (int(p_) << 11) | int(x_))
    return np.array(w, "<u2")


def evt2_words(ev):
    w = []
    hi = None
    for x_, y_, p_, t_ in zip(ev["x"], ev["y"], ev["p"], ev["t"]):
        t_ = int(t_)
        h = (t_ >> 6) & 0x0FFFFFFF
        if h != hi:
            w.append((0x8 << 28) | h)
            hi = h
        w.append(((1 if p_ else 0) << 28) | ((t_ & 0x3F) << 22) | (int(x_) << 11) | int(y_))
    return np.array(w, "<u4")


def stream(seed, n, t0, span, w=640, h=480):
    rng = np.random.default_rng(seed)
    ev = np.zeros(n, EVENT_CD)
    ev["t"] = t0 + np.sort(rng.integers(0, span, n))
    ev["x"], ev["y"], ev["p"] = rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, 2, n)
    return ev


def read_with_metavision(words, fmt, width=640, height=480):
    from metavision_core.event_io.raw_reader import RawReaderBase  # (the class the reference's iterator wraps)
    hdr = f"% evt {fmt}.0\n% format EVT{fmt};height={height};width={width}\n% geometry {width}x{height}\n% end\n".encode("ascii")
    with tempfile.NamedTemporaryFile(suffix=".raw", delete=False) as f:
        f.write(hdr)
        f.write(words.tobytes())
        path = f.name
    try:
        rd = RawReaderBase(path, delta_t=10_000)
        parts = []
        while not rd.is_done():
            e = rd.load_delta_t(-1)
            if len(e):
                parts.append(np.array(e))
        ev = np.concatenate(parts) if parts else np.zeros(0, EVENT_CD)
    finally:
        os.unlink(path)
    return {k: np.asarray(ev[k]) for k in ("x", "y", "p", "t")}


def pin_metavision():
    import metavision_sdk_base  # noqa: F401  (fails here when the SDK is absent: nothing is written)
    from metavision_sdk_cv import ActivityNoiseFilterAlgorithm
    out = {}
    cases3 = {
        "plain": evt3_words_singles(stream(1, 3000, 50_000, 40_000)),
        # words in front of the first TIME_HIGH (what a recording's first chunk looks like when it starts mid-stream)
        "no_first_time_high": evt3_words_singles(stream(2, 400, 5_000, 9_000))[1:],
        # the 24-bit time base wraps: stamps around 2^24 us
        "time_loop": evt3_words_singles(stream(3, 2000, (1 << 24) - 20_000, 40_000)),
        # TIME_HIGH repeated without a change, TIME_LOW stale behind it
        "repeated_time_high": np.concatenate((evt3_words_singles(stream(4, 50, 70_000, 3_000)), np.array([0x8000 | 17, 0x8000 | 17], "<u2"),
                                              evt3_words_singles(stream(5, 50, 17 << 12, 3_000)))),
        # vectors: VECT_BASE_X + VECT_12 + VECT_12 + VECT_8 behind a row and a time
        "vectors": np.array([0x8000 | 3, 0x6000 | 100, 0x0000 | 77, 0x3000 | (1 << 11) | 200, 0x4000 | 0xA5A, 0x4000 | 0x0F0, 0x5000 | 0x81,
                             0x6000 | 130, 0x3000 | (0 << 11) | 600, 0x4000 | 0xFFF, 0x5000 | 0x01], "<u2"),
    }
    for name, w in cases3.items():
        out[f"evt3_{name}_words"] = w
        for k, v in read_with_metavision(w, 3).items():
            out[f"evt3_{name}_{k}"] = v
    cases2 = {"plain": evt2_words(stream(6, 3000, 50_000, 40_000)),
              "time_loop": evt2_words(stream(7, 2000, (1 << 34) - 20_000, 40_000))}
    for name, w in cases2.items():
        out[f"evt2_{name}_words"] = w
        for k, v in read_with_metavision(w, 2).items():
            out[f"evt2_{name}_{k}"] = v
    # ActivityNoiseFilterAlgorithm(width, height, int(1e6 / fps)) as the reference builds it (pipe:65-67), fed positive events in
    # packets like its loop does (pipe:116-117): a dense cluster stream + isolated noise, three thresholds
    for thr in (int(1e6 / 60), 2_000, 200):
        ev = stream(10 + thr % 7, 6000, 100_000, 60_000, 64, 48)
        ev["p"] = 1
        rng = np.random.default_rng(thr)
        near = rng.random(len(ev)) < 0.6
        for i in range(1, len(ev)):
            if near[i]:
                ev["x"][i] = min(max(int(ev["x"][i - 1]) + rng.integers(-1, 2), 0), 63)
                ev["y"][i] = min(max(int(ev["y"][i - 1]) + rng.integers(-1, 2), 0), 47)
        f = ActivityNoiseFilterAlgorithm(64, 48, thr)
        buf = ActivityNoiseFilterAlgorithm.get_empty_output_buffer()
        kept = []
        cuts = list(range(0, len(ev), 700)) + [len(ev)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            f.process_events(ev[a:b], buf)
            kept.append(np.array(buf.numpy()))
        kept = np.concatenate(kept)
        for k in ("x", "y", "p", "t"):
            out[f"act{thr}_in_{k}"], out[f"act{thr}_kept_{k}"] = np.asarray(ev[k]), np.asarray(kept[k])
        out[f"act{thr}_packet_cuts"] = np.array(cuts)
    path = os.path.join(GOLDEN, "g10_metavision.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def main():
    done = 0
    for name, fn in (("cv2", pin_cv2), ("metavision", pin_metavision)):
        try:
            fn()
            done += 1
        except ImportError as e:
            print(f"{name}: not installed here ({e}); nothing written for it")
    return 0 if done else 1


if __name__ == "__main__":
    sys.exit(main())
