#!/usr/bin/env python3
"""Experiment: per-kernel HIP-event time vs events per frame (fixed overhead vs per-event slope)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S
dev = torch.device("cuda", 0)
for cfg, sizes in ((S.C_1M, (1000, 62_500, 250_000, 1_000_000, 4_000_000)), (S.C_10M, (1_000_000, 10_000_000))):
    tb = S.make_tables(cfg)
    eng = XMapsEngine(tb, n_slots=1)
    depth = torch.empty((eng.out_h, eng.out_w), dtype=torch.float32, device=dev)
    bgr = torch.empty((eng.out_h, eng.out_w, 3), dtype=torch.uint8, device=dev)
    for n in sizes:
        ev = S.make_events(cfg, n=n)
        x, y, t, _ = S.to_soa(ev)
        X, Y, T = (torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t))
        torch.cuda.synchronize()
        ms = []
        for i in range(60):
            st = eng.profile_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, n, depth.data_ptr(), bgr.data_ptr())
            ms.append(st.gpu_ms)
        m = np.array(ms)[10:].mean(0) * 1e3
        print(f"{cfg.name} n={n:>9}: minmax {m[0]:7.2f}  scatter {m[1]:7.2f}  frame {m[2]:7.2f}  total {m[3]:7.2f} us   inliers {st.n_inliers}")
    eng.close()
