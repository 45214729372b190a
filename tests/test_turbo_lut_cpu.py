"""A7's colour table: x_maps_amd/csrc/turbo_lut.inc (device) and x_maps_amd/turbo_lut.py (host / oracle) against the way OpenCV
builds COLORMAP_TURBO in its published modules/imgproc/src/colormap.cpp: Google's 256 sRGB float triplets (the same numbers
matplotlib ships as `_turbo_data`) are the knots of colormap::Turbo; init(256) runs ColorMap::linear_colormap -- a float32
piecewise-linear interpolation (slope / intercept per interval) evaluated at linspace(0, 1, 256), i.e. at the knots themselves --
and `lut.convertTo(lut, CV_8U, 255.)` rounds v * 255 to the nearest integer (cvRound: ties to even).  cv2 itself is not
available offline, so this pins the table to OpenCV's SOURCE, not to a run of it."""
import os
import re

import numpy as np
import pytest

from x_maps_amd.turbo_lut import TURBO_BGR_U8, TURBO_RGB_U8

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _google_floats():
    cm = pytest.importorskip("matplotlib._cm_listed")
    t = np.asarray(cm._turbo_data, dtype=np.float64)
    assert t.shape == (256, 3)
    return t


def _opencv_style_lut(knots_f64, interval_shift):
    """float32 slope / intercept interpolation at the knots, using the interval to the left (shift 1) or right (shift 0) of each."""
    f = np.float32
    X = np.linspace(0.0, 1.0, 256).astype(f)  # linspace<float>(0, 1, 256)
    Y = knots_f64.astype(f)                   # the static const float tables
    out = np.empty((256, 3), f)
    for k in range(256):
        i = min(max(k - interval_shift, 0), 254)
        slope = (Y[i + 1] - Y[i]) / (X[i + 1] - X[i])
        intercept = Y[i] - X[i] * slope
        out[k] = slope * X[k] + intercept
    scaled = out * f(255.0)                   # convertTo(CV_8U, 255.): float arithmetic ...
    return np.rint(scaled.astype(np.float64)).astype(np.int64)  # ... then cvRound (nearest, ties to even)


def test_table_equals_opencvs_construction_from_the_published_floats():
    g = _google_floats()
    for shift in (0, 1):
        lut = _opencv_style_lut(g, shift)
        assert lut.min() >= 0 and lut.max() <= 255
        assert np.array_equal(lut, TURBO_RGB_U8.astype(np.int64)), shift
    # no entry is anywhere near a rounding tie: float32 vs float64, interpolation noise or the rounding rule cannot matter
    frac = np.abs((g * 255.0) % 1.0 - 0.5)
    assert frac.min() > 1e-3


def test_device_table_is_the_same_table():
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{6})u", open(os.path.join(ROOT, "x_maps_amd", "csrc", "turbo_lut.inc")).read())]
    assert len(words) == 256
    dev = np.array([[w & 0xff, (w >> 8) & 0xff, (w >> 16) & 0xff] for w in words], np.uint8)  # byte0 = B, byte1 = G, byte2 = R
    assert np.array_equal(dev, TURBO_BGR_U8)
