#!/usr/bin/env python3
"""Where does DeviceIngest.poll(copy=False) spend its time?  (ESL-like stream, BGR only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
from x_maps_amd import XMapsEngine, rig, synthetic as S, _native as N
from x_maps_amd.ingest import DeviceIngest
cp, tables, evs0, _ = rig.make_esl_like(row_stride=13)
stream, _ = rig.render_stream(cp, tables, n_frames=16, row_stride=13, seed=9)
with XMapsEngine(tables) as eng:
    pin = eng.host_empty((len(stream),), S.EVENT_CD_DTYPE)
    pin[:] = stream
    packet = int(1e6 / 60 / 4)
    cuts = np.searchsorted(pin["t"], np.arange(pin["t"][0], pin["t"][-1] + packet, packet))
    for rep in range(3):
        with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 18, expected_events_per_frame=150_000, result_ring=32,
                          want_depth=False, want_bgr=True) as ing:
            for a, b in zip(cuts[:-1], cuts[1:]):
                ing.push_pinned(pin[a:b])
            ing.flush()
            fr = N.xm_ingest_frame()
            t_c, t_v, n = 0.0, 0.0, 0
            while True:
                c0 = time.perf_counter()
                rc = ing._lib.xm_ingest_poll(ing._g, C.byref(fr))
                c1 = time.perf_counter()
                if rc <= 0:
                    break
                v = ing._view(fr.bgr, (ing.shape[0], ing.shape[1], 3), C.c_uint8)
                c2 = time.perf_counter()
                t_c += c1 - c0; t_v += c2 - c1; n += 1
            print(f"rep {rep}: {n} frames; xm_ingest_poll {t_c / n * 1e6:.1f} us each, view {t_v / n * 1e6:.1f} us each")
            c0 = time.perf_counter(); s = int(v[::64, ::64].sum()); c1 = time.perf_counter()
            print(f"   touching 1/4096 of a frame: {(c1 - c0) * 1e6:.1f} us")
