#!/usr/bin/env python3
"""Experiment: where a block of the pipelined K2 (k_frame_proj_pipe) spends its items' cycles -- s_memtime between the loop's
phase marks, summed per block over its items (first 64 blocks of the LAST launch).  Needs a -DXM_ABLATE build in XM_LIB:
  XM_LIB=variants/libxmaps_abl.so python tools/k2p_phases.py [--esl] [bench flags ...]"""
import ctypes, io, os, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-other-modes", "--no-host-path", "--no-pmc", "--no-other-configs", "--groups-in-flight", "1",
            "--steps", "10", "--warmup", "2"] + sys.argv[1:]
import bench
out = io.StringIO()
with contextlib.redirect_stdout(out):
    try:
        bench.main()
    except SystemExit:
        pass
line = out.getvalue().strip().splitlines()[-1]
import json
d = json.loads(line)
print("K2 us per launch:", d["roofline"]["avg_launch_us"], "workload:", d["config"]["workload"])
from x_maps_amd import _native as N
lib = N.load_library()
buf = np.zeros((64, 16), np.uint64)
lib.xm_debug_timeline(ctypes.c_void_p(buf.ctypes.data))
names = ["0 row maxima pass (with its barriers)", "1 taps + per-disparity table", "2 barrier behind the sampling", "3 wait for the next patch + LDS writes",
         "4 output stores issued", "5 next tile record + descriptor (scalar loads)", "6 next item's loads issued", "7 barrier at the loop's end"]
items = buf[:, 8].astype(np.float64)
ok = items > 0
per = buf[ok, :8].astype(np.float64) / items[ok, None]
tot = per.sum(1).mean()
print(f"blocks {ok.sum()}, items per block {items[ok].mean():.1f}, cycles per item {tot:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:48s} {per[:, i].mean():8.0f} cycles  {100 * per[:, i].mean() / tot:5.1f} %   (min {per[:, i].min():.0f}, max {per[:, i].max():.0f})")
