#!/usr/bin/env python3
"""Experiment: per-phase s_memtime timeline of k_scatter_own_batch (owner tiles, ESL-like groups of 32; needs a -DXM_ABLATE
build in XM_LIB): thread 0 of the first 64 tiles of frame 30, one group in flight.   XM_LIB=variants/libxmaps_abl.so python tools/r05/own_timeline.py"""
import ctypes, os, sys
os.environ.setdefault("DEBUG_CLR_MAX_BATCH_SIZE", "100000")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from x_maps_amd import XMapsEngine, rig, synthetic as S, _native as N
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    N.debug_option(k, v)
B = 32
cp, tables, _, _ = rig.make_esl_like(row_stride=13, device=0)
host = [rig.render_events(cp, tables, row_stride=13, seed=f)[0] for f in range(B)]
offs = np.zeros(B + 1, np.uint64)
offs[1:] = np.cumsum([len(e) for e in host])
rec = np.empty(int(offs[-1]), S.EVENT_CD_DTYPE)
for i, e in enumerate(host):
    rec[int(offs[i]):int(offs[i + 1])] = e
dev = torch.device("cuda", 0)
aos = torch.from_numpy(rec.view(np.uint8).reshape(-1, 16).copy()).to(dev)
lib = N.load_library()
names = ["0 start", "1 locating loads issued", "2 LDS cleared, masks + thresholds in place (barrier)", "3 events arrived, columns found",
         "4 LUT gathers issued", "5 both gathers arrived, slots worked out", "6 pass 0: ds_max done, barrier passed",
         "7 every row pass flushed", "8 extras flushed"]
acc = []
with XMapsEngine(tables, device=0, n_slots=B) as eng:
    depth = torch.zeros((B, eng.out_h, eng.out_w), dtype=torch.float32, device=dev)
    print(eng.cols_info())
    for it in range(10):
        eng.process_events_batch_device(aos.data_ptr(), offs, depth.data_ptr(), None)
        eng.sync()
        buf = np.zeros((64, 16), np.uint64)
        lib.xm_debug_timeline(ctypes.c_void_p(buf.ctypes.data))
        if it >= 2:
            acc.append(buf[:, :9].astype(np.int64) - buf[:, :1].astype(np.int64))
    print(eng.path_counts())
a = np.mean(acc, axis=0) * 10.0  # ns (s_memtime: 100 MHz)
a = a[1:63]  # (tile 0 holds the rig's extras: not typical)
prev = 0
for i, nm in enumerate(names):
    m = a[:, i].mean()
    print(f"{nm:56s} t={m:9.0f} ns  (+{m - prev:8.0f})   p10 {np.percentile(a[:, i], 10):8.0f}  p90 {np.percentile(a[:, i], 90):8.0f}")
    prev = m
