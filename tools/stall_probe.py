#!/usr/bin/env python3
"""Experiment: per-call host latency of the frame loop -> where are the periodic multi-millisecond stalls?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S
dev = torch.device("cuda", 0)
cfg = S.C_1M
tb = S.make_tables(cfg)
slots = int(os.environ.get("SLOTS", "4"))
eng = XMapsEngine(tb, n_slots=slots, try_sorted=os.environ.get("MODE") == "try")
frames = []
for f in range(8):
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f))
    frames.append(tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)))
out = torch.empty((slots, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
bgr = torch.empty((slots, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
N = 8000
ts = np.zeros(N + 1)
ts[0] = time.perf_counter()
for i in range(N):
    fx, fy, ft = frames[i % 8]
    eng.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, cfg.n_events, out[i % slots].data_ptr(), bgr[i % slots].data_ptr())
    ts[i + 1] = time.perf_counter()
eng.sync()
d = np.diff(ts) * 1e6
big = np.nonzero(d > 100)[0]
print(f"calls {N}: median {np.median(d):.2f} us, mean {d.mean():.2f}, calls > 100 us: {len(big)}")
print("  at call index (latency us): " + "  ".join(f"{i}({d[i]:.0f})" for i in big[:40]))
if len(big) > 2: print("  spacing between them: " + " ".join(str(int(v)) for v in np.diff(big)[:40]))
