#!/usr/bin/env python3
"""Experiment: per-phase s_memtime timeline of k_frame_proj_tiled_batch<2> at full occupancy (60-frame launches; needs a
-DXM_ABLATE build in XM_LIB): thread 0 of 40 tiles of row 15 of frame 30.   python tools/k2_timeline.py"""
import ctypes, os, sys
os.environ.setdefault("DEBUG_CLR_MAX_BATCH_SIZE", "100000")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S, _native as N
F = 60
cfg = S.C_1M
tb = S.make_tables(cfg)
dev = torch.device("cuda", 0)
n = cfg.n_events
X = torch.empty(F * n, dtype=torch.int16, device=dev); Y = torch.empty_like(X); T = torch.empty(F * n, dtype=torch.int64, device=dev)
for f in range(F):
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f % 8))
    X[f * n:(f + 1) * n] = torch.from_numpy(x.view(np.int16)); Y[f * n:(f + 1) * n] = torch.from_numpy(y.view(np.int16)); T[f * n:(f + 1) * n] = torch.from_numpy(t)
depth = torch.zeros((F, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
bgr = torch.zeros((F, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
offs = np.arange(F + 1, dtype=np.uint64) * n
lib = N.load_library()
names = ["0 start", "1 patch loaded + stored to LDS", "2 barrier 1", "3 7-tap column maxima", "4 barrier 2", "5 7 taps per pixel",
         "6 table lookup, depth + BGR stored"]
acc = []
with XMapsEngine(tb, n_slots=F) as eng:
    for it in range(8):
        eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(), bgr.data_ptr())
        eng.sync()
        buf = np.zeros((64, 16), np.uint64)
        lib.xm_debug_timeline(ctypes.c_void_p(buf.ctypes.data))
        if it >= 2:
            acc.append(buf[:40, 9:16].astype(np.int64) - buf[:40, 9:10].astype(np.int64))
a = np.mean(acc, axis=0)
prev = 0
tot = a[:, 6].mean()
for i, nm in enumerate(names):
    m = a[:, i].mean()
    print(f"{nm:44s} t={m:9.0f} ticks  (+{m - prev:8.0f} = {100 * (m - prev) / tot:5.1f} %)")
    prev = m
