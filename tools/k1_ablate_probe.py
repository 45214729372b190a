#!/usr/bin/env python3
"""Experiment: what bounds the one-tile K1 (k_scatter_cols_batch) at full occupancy?  32-frame groups at C-1M through
xm_profile_batch (serial launches, per-launch HIP events) with parts of the kernel switched off (-DXM_ABLATE build in XM_LIB,
XM_ABLATE=<mask>: 16 no flush stores, 32 no band loads, 64 no event loads, 128 no per-event work).  Results are wrong by
construction; only the launch durations matter.   XM_ABLATE=48 python tools/k1_ablate_probe.py"""
import os, sys
os.environ.setdefault("DEBUG_CLR_MAX_BATCH_SIZE", "100000")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S
F, G = 32, 3
cfg = S.C_1M
tb = S.make_tables(cfg)
dev = torch.device("cuda", 0)
n = cfg.n_events
X = torch.empty(G * F * n, dtype=torch.int16, device=dev); Y = torch.empty_like(X); T = torch.empty(G * F * n, dtype=torch.int64, device=dev)
for f in range(G * F):
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f % 16))
    X[f * n:(f + 1) * n] = torch.from_numpy(x.view(np.int16)); Y[f * n:(f + 1) * n] = torch.from_numpy(y.view(np.int16)); T[f * n:(f + 1) * n] = torch.from_numpy(t)
depth = torch.zeros((F, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
offs = [np.arange(F + 1, dtype=np.uint64) * n + g * F * n for g in range(G)]
res = []
with XMapsEngine(tb, n_slots=F) as eng:
    for it in range(30):
        g = it % G
        ms = eng.profile_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs[g], depth.data_ptr(), None)
        if it >= 9:
            res.append(ms)
    pc = eng.path_counts()
a = np.median(np.array(res), axis=0) * 1000.0 / F
print("XM_ABLATE=%s  us/frame: K0b %.2f  K1 %.2f  K2 %.2f   paths %s" % (os.environ.get("XM_ABLATE", "0"), a[0], a[1], a[2], pc))
