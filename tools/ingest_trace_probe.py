#!/usr/bin/env python3
"""ESL-like stream through the device ingest (BGR views): where the launch side's time goes (XM_INGEST_TRACE) + rates"""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_TORCH") == "1":  # (PyTorch's bundled HIP runtime serves the process, as in bench.py)
    import torch
    torch.cuda.init()
import numpy as np
from x_maps_amd import XMapsEngine, rig, synthetic as S
from x_maps_amd.ingest import DeviceIngest
from x_maps_amd import _native as _N
_N.debug_option("XM_INGEST_TRACE", "1")
cp, tables, evs0, _ = rig.make_esl_like(row_stride=13)
stream, _ = rig.render_stream(cp, tables, n_frames=48, row_stride=13, seed=9)
WANT_BGR = os.environ.get("PROBE_NO_OUT") != "1"  # (PROBE_NO_OUT=1: no result frames cross PCIe -- what is left is the ingest itself)
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
mp = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
with XMapsEngine(tables) as eng:
    pin = eng.host_empty((len(stream),), S.EVENT_CD_DTYPE)
    pin[:] = stream
    packet = int(1e6 / 60 / 4)
    cuts = np.searchsorted(pin["t"], np.arange(pin["t"][0], pin["t"][-1] + packet, packet))
    for rep in range(3):
        with DeviceIngest(eng, 60, capacity_events=cap, max_packet_events=mp, result_ring=64, want_depth=False, want_bgr=WANT_BGR) as ing:
            nw = int(os.environ.get("PROBE_WARM", "4"))
            for a, b in zip(cuts[:nw], cuts[1:nw + 1]):
                ing.push_pinned(pin[a:b])
            ing.flush(), ing.reset(), ing.poll(copy=False)
            c0 = time.perf_counter()
            for a, b in zip(cuts[:-1], cuts[1:]):
                ing.push_pinned(pin[a:b])
            c1 = time.perf_counter()
            ing.flush()
            got = ing.poll(copy=False)
            dt = time.perf_counter() - c0
            print(f"cap {cap} max_packet {mp}: {len(got)} frames, {len(cuts) - 1} pushes in {dt * 1e3:.2f} ms (push loop {(c1 - c0) * 1e3:.2f}) = {len(stream) / dt / 1e6:.1f} Mev/s", flush=True)
