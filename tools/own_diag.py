#!/usr/bin/env python3
"""Diagnose a failing seed of tests/test_gpu_own.py::test_random_shared_cell_rigs_and_streams at the level of A3's output:
the u16 disparity frame of the tiles (xm_debug_last_disp_frame) against the oracle's disp_map, cell by cell.
  python tools/own_diag.py first_seed n_seeds [max_reports]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["XM_COLS"] = "2"
import numpy as np
import test_gpu_own as W
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250
max_rep = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reports = 0
for seed in range(first, first + n):
    try:
        tb, evs = W._random_shared_rig(seed)
    except ValueError:
        continue
    ref = W._ref(tb, evs)
    with XMapsEngine(tb, n_slots=4) as eng:
        info = eng.cols_info()
        got = W._run(eng, evs)
        ok = W._same(got, ref)
        pc = eng.path_counts()
        fb = eng.sorted_fallbacks()
        if ok:
            continue
        reports += 1
        d, b, st = got
        print(f"== seed {seed}: info={info} paths={pc} fallbacks={fb} n={len(evs)} inliers gpu={st.n_inliers} ref={int(ref['mask'].sum())} "
              f"depth diffs={int((d != ref['depth']).sum())}", flush=True)
        if info["mode"] == "none" or fb:
            continue
        f16 = eng.debug_last_disp_frame().astype(np.int64)
        dm = np.asarray(ref["disp_map"]).astype(np.int64)
        bad = np.argwhere(f16 != dm)
        print(f"   cells that differ: {len(bad)}")
        X = tb["proj_x_map"].astype(np.int64)
        xo = tb["x_offset"]
        x_, y_, t_, _ = S.to_soa(evs)
        ts = np.rint(((t_ - t_.min()) / (t_.max() - t_.min())) * tb["t_px_scale"]).astype(np.int64)
        for (r, x) in bad[:6]:
            cols = np.nonzero(X[r] - xo == x)[0]
            # the events of the oracle that hit the cell
            hit = np.nonzero(ref["mask"] & (ref["yr"] == r) & ((ref["xr"].astype(np.int64) + ref["disp"]) == x))[0]
            print(f"   cell row {r} x {x}: gpu {f16[r, x]} ref {dm[r, x]}; X-map columns of the row on this cell: {cols.tolist()} "
                  f"(tiles {sorted(set((cols // info['w']).tolist())) if info.get('w') else '-'}); events on it: {hit[-4:].tolist()} "
                  f"cols {ts[hit[-4:]].tolist()} disp {ref['disp'][hit[-4:]].tolist()}")
            lo, hi = max(0, (cols.min() if len(cols) else 0) - 10), (cols.max() if len(cols) else 0) + 10
            print(f"      row's X-map - x_offset around: c{lo}.. {(X[r, lo:hi] - xo).tolist()}")
    if reports >= max_rep:
        break
print("reports:", reports)
