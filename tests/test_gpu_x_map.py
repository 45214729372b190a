"""-m gpu: N1, X-map construction on the GPU vs the reference's output (golden G3) and the oracle."""
import os
import time

import numpy as np
import pytest

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


def test_linear_time_map_host_restatement(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_time_map.npz"))
    for key in g.files:
        _, wh, d = key.split("_")
        w, h = map(int, wh.split("x"))
        assert np.array_equal(generate_linear_projector_time_map(w, h, d == "up"), g[key])
