#!/usr/bin/env python3
"""Experiment: per-phase s_memtime timeline of k_scatter_tiled (needs the -DXM_ABLATE build, see tools/timeline.sh)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_maps_amd import XMapsEngine, synthetic as S, _native as N
dev = torch.device("cuda", 0)
cfg = S.C_1M
tb = S.make_tables(cfg)
eng = XMapsEngine(tb, n_slots=1)
ev = S.make_events(cfg)
x, y, t, _ = S.to_soa(ev)
X, Y, T = (torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t))
depth = torch.empty((eng.out_h, eng.out_w), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
lib = N.load_library()
names = ["0 start", "1 event+sample loads issued", "2 extrema loaded, TimeNorm", "3 window from samples", "4 bands loaded+stored, slots zeroed",
         "5 time columns computed", "6 barrier1", "7 fast + slow pass done", "8 barrier2", "9 flush issued", "10 end",
         "11 (bands stored to LDS)", "12 (slow gather 2 issued, before barrier1)",
         "13 (LUT + X-map gathers consumed, before the slot-reuse barrier)", "14 (slots cleared, second barrier passed)"]
acc = []
acc_raw = []
for it in range(30):
    eng.process_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, len(t), depth.data_ptr(), None)
    eng.sync()
    buf = np.zeros((64, 16), np.uint64)
    lib.xm_debug_timeline(ctypes.c_void_p(buf.ctypes.data))
    if it >= 5:
        acc_raw.append(buf.astype(np.float64))
        acc.append((buf[:, :15].astype(np.int64) - buf[:, :1].astype(np.int64)))
a = np.mean(acc, axis=0)  # [block][phase] in s_memtime ticks (100 MHz constant clock on gfx9: 10 ns)
print("phase                          mean over blocks 0..63   (s_memtime ticks; scale with the kernel duration printed below)")
prev = 0
for i, nm in enumerate(names):
    m = a[:, i].mean()
    print(f"{nm:44s} t={m:9.1f}  (+{m - prev:8.1f})" if i < 11 else f"{nm:44s} t={m:9.1f}")
    prev = m
import subprocess
print("kernel duration by events:", eng.profile_frame_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, len(t), depth.data_ptr(), None).gpu_ms)
