#!/usr/bin/env python3
"""Experiment: is the pipelined frame rate limited by the single host thread that enqueues?  Drive T engines from T threads."""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S
dev = torch.device("cuda", 0)
cfg = S.C_1M
tb = S.make_tables(cfg)
frames = []
for f in range(8):
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f))
    frames.append(tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)))
torch.cuda.synchronize()
K = 400
for T in (1, 2, 3):
    for slots in (4, 8):
        engs = [XMapsEngine(tb, n_slots=slots) for _ in range(T)]
        outs = [torch.empty((slots, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev) for _ in range(T)]
        bgrs = [torch.empty((slots, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev) for _ in range(T)]
        torch.cuda.synchronize()
        def work(k, n):
            e = engs[k]
            for i in range(n):
                fx, fy, ft = frames[(i + k) % 8]
                e.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, cfg.n_events,
                                       outs[k][i % slots].data_ptr(), bgrs[k][i % slots].data_ptr())
            e.sync()
        for k in range(T): work(k, 40)
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k, K)) for k in range(T)]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
        print(f"threads {T} x slots {slots}: {T*K*cfg.n_events/dt/1e6:9.0f} Mev/s  ({dt/(T*K)*1e6:.2f} us/frame)")
        for e in engs: e.close()
