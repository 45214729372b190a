"""-m gpu: N1, X-map construction on the GPU vs the reference's output (golden G3) and the oracle."""
import os
import time

import numpy as np
import pytest

from conftest import xm_option

import xmaps_oracle as O
from x_maps_amd.proj_time_map import generate_linear_projector_time_map
from x_maps_amd.x_map import compute_x_map_from_time_map

pytestmark = pytest.mark.gpu


def test_x_map_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_x_map.npz"))
    xm, td = compute_x_map_from_time_map(g["time_map"], int(g["x_map_width"]), int(g["t_px_scale"]), 4242,
                                         int(g["num_scanlines"]))
    assert xm.dtype == np.int16 and np.array_equal(xm, g["x_map"])
    assert np.allclose(td, g["t_diffs"], rtol=0, atol=2e-7)  # by-product: stub run is f32, Numba/GPU f64 (see oracle test)


def test_x_map_matches_oracle_at_esl_like_size():
    """Rectified linear time map with holes, 330 x 440 rows/cols, 270 time columns (C-ESL / 4)."""
    rng = np.random.default_rng(2)
    h, w, tw = 330, 440, 270
    tm = generate_linear_projector_time_map(w, h, True)
    tm = (tm + rng.normal(0, 1e-4, tm.shape)).astype(np.float32)
    tm[:7] = 0
    tm[-5:] = 0
    tm[:, :9] = 0
    tm[rng.random(tm.shape) < 0.02] = 0
    t0 = time.perf_counter()
    ref_x, ref_d = O.compute_x_map_from_time_map(tm, tw, tw - 1, 4242, w)
    t1 = time.perf_counter()
    xm, td = compute_x_map_from_time_map(tm, tw, tw - 1, 4242, w)
    t2 = time.perf_counter()
    assert np.array_equal(xm, ref_x) and np.array_equal(td, ref_d)
    assert (xm[:, 0] == 0).all() and (xm > 4242).mean() > 0.5
    print(f"x-map {h}x{tw} from {h}x{w}: oracle {1e3 * (t1 - t0):.1f} ms, GPU call {1e3 * (t2 - t1):.1f} ms")


def _adversarial_time_map(seed):
    """rows that are not monotone, with duplicates, exact ties around a time column's t, undefined cells, values far below t,
    negative values, rows without a defined cell, a NaN"""
    rng = np.random.default_rng(500 + seed)
    h, w, tw = int(rng.integers(8, 24)), int(rng.integers(3, 420)), int(rng.integers(2, 300))
    S_ = tw - 1 if tw > 1 else 1
    tm = rng.random((h, w)).astype(np.float32)
    kind = rng.integers(0, 5, h)
    for r in range(h):
        if kind[r] == 0:  # a smooth ramp with noise (the usual case)
            tm[r] = (np.linspace(0.02, 0.98, w) + rng.normal(0, 2e-3, w)).astype(np.float32)
        elif kind[r] == 1:  # few distinct values: long runs of duplicates
            tm[r] = rng.choice(np.float32(rng.random(5)), w)
        elif kind[r] == 2:  # exact ties: t +- d for time columns' t, in random order
            c = rng.integers(1, max(tw, 2), w)
            d = rng.choice(np.float64([0.0, 2.0 ** -12, 2.0 ** -10, 3 * 2.0 ** -11]), w)
            sgn = rng.choice([-1.0, 1.0], w)
            tm[r] = (np.float32(c / S_) + np.float32(sgn * d)).astype(np.float32)
        elif kind[r] == 3:  # values many orders of magnitude below t: their distances collapse in float64
            tm[r] = (rng.random(w) * 1e-30).astype(np.float32)
            tm[r, rng.integers(0, w)] = np.float32(0.5)
        # kind 4: uniform random
    tm[rng.random(tm.shape) < 0.1] = 0
    if h > 2:
        tm[1] = 0  # a row without a defined cell
    if seed % 3 == 0:
        tm[rng.integers(0, h), rng.integers(0, w)] = -0.25
    if seed % 7 == 0:
        tm[0, w // 2] = np.nan
    return tm, tw, S_, int(rng.integers(1, 400))


@pytest.mark.parametrize("seed", range(24))
def test_sorted_row_builder_equals_the_exhaustive_scan_and_the_oracle(monkeypatch, seed):
    """N1's kernel sorts every row once and finds a time column's x by binary search (k_build_x_map_sorted); the exhaustive scan
    (the reference's loop, python/x_map.py:26-52) stays as XM_XMAP_SCAN=1: both must agree bit for bit, with the oracle too."""
    tm, tw, S_, nsl = _adversarial_time_map(seed)
    xm, td = compute_x_map_from_time_map(tm, tw, S_, 4242, nsl)
    xm_option("XM_XMAP_SCAN", "1")
    xm_s, td_s = compute_x_map_from_time_map(tm, tw, S_, 4242, nsl)
    assert np.array_equal(xm, xm_s) and np.array_equal(td, td_s), seed
    if not np.isnan(tm).any():
        ref_x, ref_d = O.compute_x_map_from_time_map(tm, tw, S_, 4242, nsl)
        assert np.array_equal(xm, ref_x) and np.array_equal(td, ref_d), seed


def test_linear_time_map_host_restatement(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_time_map.npz"))
    for key in g.files:
        _, wh, d = key.split("_")
        w, h = map(int, wh.split("x"))
        assert np.array_equal(generate_linear_projector_time_map(w, h, d == "up"), g[key])
