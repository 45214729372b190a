"""-m gpu: a host that is neither Python nor C++.  tests/c_host/shard_host.c (plain C99, include/xmaps.h only, linked with
libxmaps_hip.so and nothing else) runs one C-1M frame through the library's shard communicator (world of one rank: columns merge
and packed keys), through xm_create_sharded and through xm_process_frame -- in a process without torch, i.e. on ROCm's own HIP
runtime with the librccl the library finds itself -- and every frame it writes back equals the oracle's."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from c_oracle import COracle
from x_maps_amd import _native as N
from x_maps_amd import synthetic as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_runs_the_sharded_entries(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    N.build_native()
    exe = tmp_path / "shard_host"
    libdir = os.path.dirname(N.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-O1", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_host", "shard_host.c"), "-o", str(exe), "-L", libdir, "-lxmaps_hip",
                    "-Wl,-rpath," + libdir], check=True)
    cfg = S.C_1M
    tb = S.make_tables(cfg)
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=7))
    ref = COracle(tb, False, omp=True).process_ev_frame(x, y, t)
    mapx, pmap, xmap = np.asarray(tb["cam_mapx_i16"]), np.asarray(tb["disp_proj_mapxy_i16"]), np.asarray(tb["proj_x_map"])
    hd = np.zeros(16, np.int32)
    hd[:10] = [mapx.shape[1], mapx.shape[0], pmap.shape[1], pmap.shape[0], tb["rect_w"], tb["rect_h"], xmap.shape[1], xmap.shape[0],
               tb.get("x_offset", 4242), len(t)]
    blob = tmp_path / "frame.blob"
    with open(blob, "wb") as f:
        f.write(hd.tobytes())
        f.write(np.float64(tb["p03"]).tobytes())
        f.write(np.array([tb["z_near"], tb["z_far"]], np.float32).tobytes())
        for a in (tb["cam_mapx_i16"], tb["cam_mapy_i16"], tb["proj_x_map"], tb["disp_proj_mapxy_i16"]):
            f.write(np.ascontiguousarray(a, dtype=np.int16).tobytes())
        f.write(np.ascontiguousarray(x, dtype=np.uint16).tobytes())
        f.write(np.ascontiguousarray(y, dtype=np.uint16).tobytes())
        f.write(np.ascontiguousarray(t, dtype=np.int64).tobytes())
    r = subprocess.run([str(exe), str(blob), str(tmp_path / "out")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert "comm: columns=1" in r.stdout and "sharded: n_dev=1 rccl=1 columns=1 keys=0 redone=0" in r.stdout, r.stdout
    px = pmap.shape[0] * pmap.shape[1]
    for w in (2, 4, 8):
        assert f"virtual: n_dev={w} rccl=0 columns=1 keys=0 redone=0" in r.stdout, r.stdout
    for leg in ("comm_columns", "comm_keys", "sharded", "sharded_w2", "sharded_w4", "sharded_w8", "single"):
        raw = np.fromfile(tmp_path / f"out.{leg}", dtype=np.uint8)
        depth = raw[:px * 4].view(np.float32).reshape(pmap.shape[:2])
        bgr = raw[px * 4:].reshape(pmap.shape[:2] + (3,))
        assert np.array_equal(depth, ref["depth"]) and np.array_equal(bgr, ref["bgr"]), leg
