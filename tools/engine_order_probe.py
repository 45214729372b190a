#!/usr/bin/env python3
"""Experiment: the same pipelined loop on engines created at different points of the process (before / after the event
buffers are uploaded, first / second engine).  bench.py saw the second engine run 14 % faster than the first."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S
dev = torch.device("cuda", 0)
cfg = S.C_1M
tb = S.make_tables(cfg)
slots = int(os.environ.get('SLOTS', '4'))
order = os.environ.get("ORDER", "AB-upload-C")
engs = {}
def mk(name): engs[name] = XMapsEngine(tb, n_slots=slots, try_sorted=bool(os.environ.get('TRY')))
frames = []
def upload():
    for f in range(8):
        x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f))
        frames.append(tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)))
for tok in order.split("-"):
    if tok == "upload": upload()
    elif tok == "x":  # a throwaway engine, created and closed
        XMapsEngine(tb, n_slots=slots).close()
    elif tok == "m":  # throwaway allocations only
        junk = [torch.empty(20_000_000, dtype=torch.uint8, device=dev) for _ in range(12)]
        torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
    elif tok.startswith("d"):  # dN: N placeholder HIP streams created through torch and kept alive, never used
        keep = globals().setdefault("keep_streams", [])
        keep += [torch.cuda.Stream() for _ in range(int(tok[1:]))]
    elif tok == "s":  # throwaway streams only
        ss = [torch.cuda.Stream() for _ in range(8)]
        for q in ss:
            with torch.cuda.stream(q): torch.zeros(8, device=dev)
        torch.cuda.synchronize()
    else:
        for c in tok: mk(c)
out = torch.empty((slots, cfg.proj_h, cfg.proj_w), dtype=torch.float32, device=dev)
bgr = torch.empty((slots, cfg.proj_h, cfg.proj_w, 3), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def run(e, n):
    for i in range(n):
        fx, fy, ft = frames[i % 8]
        e.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, cfg.n_events, out[i % slots].data_ptr(), bgr[i % slots].data_ptr())
    e.sync()
for rep in range(2):
    for name, e in engs.items():
        run(e, 40)
        t0 = time.perf_counter(); run(e, 400); dt = time.perf_counter() - t0
        print(f"order {order} pass {rep} engine {name}: {400*cfg.n_events/dt/1e6:9.0f} Mev/s ({dt/400*1e6:.2f} us/frame)  key_frame[0] @ {e.stream(0):#x}")
