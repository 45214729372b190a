#!/usr/bin/env python3
"""Experiment: is the pipelined frame rate steady over a long run?  Times every chunk of 200 frames (sync at chunk ends)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S
dev = torch.device("cuda", 0)
cfg = S.C_1M
tb = S.make_tables(cfg)
slots = int(os.environ.get("SLOTS", "4"))
mode = os.environ.get("MODE", "general")
eng = XMapsEngine(tb, n_slots=slots, try_sorted=mode == "try", assume_time_sorted=mode == "sorted")
frames = []
for f in range(8):
    x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f))
    frames.append(tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)))
out = torch.empty((slots, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
bgr = torch.empty((slots, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def run(n, i0=0):
    for i in range(i0, i0 + n):
        fx, fy, ft = frames[i % 8]
        eng.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, cfg.n_events, out[i % slots].data_ptr(), bgr[i % slots].data_ptr())
chunk = int(os.environ.get("CHUNK", "200"))
sync_each = os.environ.get("SYNC_EACH", "1") == "1"
run(40); eng.sync()
if os.environ.get("NOGC"):
    import gc; gc.disable()
res = []
T0 = time.perf_counter()
for c in range(40):
    t0 = time.perf_counter(); run(chunk, c * chunk)
    if sync_each: eng.sync()
    res.append((time.perf_counter() - t0) / chunk * 1e6)
eng.sync()
tot = (time.perf_counter() - T0) / (40 * chunk) * 1e6
print(f"mode {mode} slots {slots} chunk {chunk} sync_each {sync_each}: overall {tot:.2f} us/frame = {cfg.n_events/tot:.0f} Mev/s")
print("  us/frame per chunk: " + " ".join(f"{r:.1f}" for r in res))
