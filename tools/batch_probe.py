#!/usr/bin/env python3
"""Minimal driver of the multi-frame launches for profilers: F x C-1M frames in one grid per kernel, REPS launches.
    python tools/batch_probe.py [F=60] [REPS=5] [mode: 0 general (K0), 1 declared sorted, 2 library defaults]"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_MAX_BATCH_SIZE", "100000")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_maps_amd import XMapsEngine
from x_maps_amd import synthetic as S

F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
SORTED = len(sys.argv) > 3 and sys.argv[3] == "1"
DEFAULT = len(sys.argv) > 3 and sys.argv[3] == "2"  # library defaults: verified shortcut + compact key frame
cfg = S.C_1M
tb = S.make_tables(cfg)
dev = torch.device("cuda", 0)
n = cfg.n_events
X = torch.empty(F * n, dtype=torch.int16, device=dev)
Y = torch.empty_like(X)
T = torch.empty(F * n, dtype=torch.int64, device=dev)
for f in range(F):
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f % 8))
    X[f * n:(f + 1) * n] = torch.from_numpy(x.view(np.int16))
    Y[f * n:(f + 1) * n] = torch.from_numpy(y.view(np.int16))
    T[f * n:(f + 1) * n] = torch.from_numpy(t)
depth = torch.zeros((F, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
bgr = torch.zeros((F, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
offs = np.arange(F + 1, dtype=np.uint64) * n
with XMapsEngine(tb, n_slots=F, force_general=not (SORTED or DEFAULT), assume_time_sorted=SORTED) as eng:
    for _ in range(REPS):
        eng.process_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(), bgr.data_ptr())
        eng.sync()
print("done", F, REPS)
