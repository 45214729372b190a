#!/usr/bin/env python3
"""Experiment: per-phase s_memtime timeline of k_scatter_cols_batch at full occupancy (60-frame launches; needs a -DXM_ABLATE
build in XM_LIB): thread 0 of the first 64 tiles of frame 30.   python tools/cols_timeline.py"""
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
torch.cuda.synchronize()
offs = np.arange(F + 1, dtype=np.uint64) * n
lib = N.load_library()
names = ["0 start", "1 bounds / thresholds / t0,tn / tag arrived", "2 events arrived", "3 event arithmetic done", "4 bands arrived",
         "5 barrier 1 passed", "6 gathers + ds_max done", "7 barrier 2 passed", "8 flush issued"]
acc = []
with XMapsEngine(tb, n_slots=F) as eng:
    for it in range(8):
        eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(), None)
        eng.sync()
        buf = np.zeros((64, 16), np.uint64)
        lib.xm_debug_timeline(ctypes.c_void_p(buf.ctypes.data))
        if it >= 2:
            acc.append(buf[:, :9].astype(np.int64) - buf[:, :1].astype(np.int64))
    print(eng.path_counts())
a = np.mean(acc, axis=0) * 10.0  # ns (s_memtime: 100 MHz)
prev = 0
for i, nm in enumerate(names):
    m = a[:, i].mean()
    print(f"{nm:48s} t={m:9.0f} ns  (+{m - prev:8.0f})   p10 {np.percentile(a[:, i], 10):8.0f}  p90 {np.percentile(a[:, i], 90):8.0f}")
    prev = m
