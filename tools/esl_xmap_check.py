import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import rig
cp, tb, _, _ = rig.make_esl_like(row_stride=13, device=0)
X = np.asarray(tb["proj_x_map"]); print("xmap", X.shape, X.dtype, "rect", tb["rect_w"], tb["rect_h"], "x_offset", tb.get("x_offset"))
mx = np.asarray(tb["cam_mapx_i16"]); xr_min = int(mx.min()); print("xr_min", xr_min, "xr_max", mx.max(), "xp", X.min(), X.max())
xo = int(tb.get("x_offset", 4242))
rows = min(X.shape[0]-1, int(tb["rect_h"]))
fu = X[:rows].astype(np.int64) - xo
live = fu >= xr_min
print("live frac", live.mean())
dup = 0; rowsdup = 0; gaps = []
for r in range(rows):
    idx = np.nonzero(live[r])[0]; v = fu[r][idx]
    o = np.argsort(v, kind="stable"); vs = v[o]; same = vs[1:] == vs[:-1]
    d = int(same.sum()); dup += d; rowsdup += d > 0
    if d: gaps.append(np.abs(idx[o][1:][same] - idx[o][:-1][same]))
print("dups", dup, "rows with dups", rowsdup, "of", rows)
if gaps:
    g = np.concatenate(gaps); print("column distance of duplicate pairs: min", g.min(), "max", g.max(), np.bincount(np.minimum(g, 10)))
r = rows//2; v = fu[r]; print(v[:12], v[500:512], v[-12:])
