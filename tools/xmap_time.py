import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from x_maps_amd.proj_time_map import generate_linear_projector_time_map
from x_maps_amd.x_map import compute_x_map_from_time_map
rng = np.random.default_rng(1)
h, w, tw = 1320, 1760, 1080
tm = generate_linear_projector_time_map(w, h, True)
tm = (tm + rng.normal(0, 1e-4, tm.shape)).astype(np.float32)
tm[rng.random(tm.shape) < 0.02] = 0
for env in ("0", "1"):
    os.environ["XM_XMAP_SCAN"] = env
    compute_x_map_from_time_map(tm, tw, tw - 1, 4242, w)
    t0 = time.perf_counter()
    for _ in range(5):
        r = compute_x_map_from_time_map(tm, tw, tw - 1, 4242, w)
    print("XM_XMAP_SCAN=%s: %.2f ms per call (incl. copies)" % (env, (time.perf_counter() - t0) / 5 * 1e3))
    if env == "0": a = r
    else: print("equal:", np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1]))
