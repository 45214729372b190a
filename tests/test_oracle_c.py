"""CPU: the C restatement (oracle/xmaps_oracle.c, serial and OpenMP) == the NumPy oracle == the golden vectors."""
import os

import numpy as np
import pytest

import xmaps_oracle as O
from c_oracle import COracle
from x_maps_amd import synthetic as S


@pytest.mark.parametrize("omp", [False, True])
@pytest.mark.parametrize("camera", [False, True])
def test_c_oracle_equals_numpy_oracle(omp, camera):
    tb = S.make_tables(S.C_TINY)
    co = COracle(tb, camera, omp=omp)
    for frame, kw in ((0, {}), (1, {"shuffled": True}), (2, {"n": 17})):
        evs = S.make_events(S.C_TINY, frame=frame, **kw)
        x, y, t, _ = S.to_soa(evs)
        ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera)
        got = co.process_ev_frame(x, y, t)
        assert np.array_equal(got["mask"], ref["mask"]) and np.array_equal(got["disp"], ref["disp"])
        assert np.array_equal(got["disp_map"], ref["disp_map"])
        assert np.array_equal(got["depth"], ref["depth"]) and np.array_equal(got["bgr"], ref["bgr"])


@pytest.mark.parametrize("name", ["g1a_n1000", "g1b_n100000", "g1c_unsorted_dups", "g1d_rint_ties", "g1e_edges", "g1h_equal_t"])
def test_c_oracle_against_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    tb = {"cam_mapx_i16": g["mapx"], "cam_mapy_i16": g["mapy"], "proj_x_map": g["xmap"],
          "disp_proj_mapxy_i16": np.zeros((2, 2, 2), np.int16), "rect_w": int(g["rect_w"]), "rect_h": int(g["rect_h"]),
          "p03": 1.0, "z_near": 0.1, "z_far": 1.2}
    got = COracle(tb, False, omp=True).process_ev_frame(g["x"], g["y"], g["t"])
    assert np.array_equal(got["mask"], g["mask"]) and np.array_equal(got["disp"], g["disp"])
    assert np.array_equal(got["disp_map"], g["disp_map_proj"])
    gotc = COracle(tb, True).process_ev_frame(g["x"], g["y"], g["t"])
    assert np.array_equal(gotc["disp_map"], g["disp_map_cam"])
