#!/usr/bin/env python3
"""Experiment: pipelined throughput of the AoS EventCD input (what Metavision hands over) vs the SoA columns, device-resident."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S
dev = torch.device("cuda", 0)
cfg = S.C_1M
tb = S.make_tables(cfg)
slots = 8
for sorted_mode in (False, True):
    eng = XMapsEngine(tb, n_slots=slots, assume_time_sorted=sorted_mode)
    aos, soa = [], []
    for f in range(8):
        ev = S.make_events(cfg, frame=f)
        aos.append(torch.from_numpy(ev.view(np.uint8).reshape(-1)).to(dev))
        x, y, t, _ = S.to_soa(ev)
        soa.append(tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)))
    out = torch.empty((slots, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
    bgr = torch.empty((slots, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    def run_aos(n, use_p):
        for i in range(n):
            eng.process_events_device(aos[i % 8].data_ptr(), cfg.n_events, use_polarity=use_p, depth_ptr=out[i % slots].data_ptr(), bgr_ptr=bgr[i % slots].data_ptr())
        eng.sync()
    def run_soa(n):
        for i in range(n):
            fx, fy, ft = soa[i % 8]
            eng.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, cfg.n_events, out[i % slots].data_ptr(), bgr[i % slots].data_ptr())
        eng.sync()
    for name, fn in (("SoA", lambda n: run_soa(n)), ("AoS", lambda n: run_aos(n, False)), ("AoS + polarity mask", lambda n: run_aos(n, True))):
        fn(40)
        t0 = time.perf_counter(); fn(400); dt = time.perf_counter() - t0
        print(f"sorted={sorted_mode} {name:20s}: {400*cfg.n_events/dt/1e6:9.0f} Mev/s ({dt/400*1e6:.2f} us/frame)")
    eng.close()
