#!/usr/bin/env python3
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from x_maps_amd import XMapsEngine, rig, synthetic as S
from x_maps_amd.ingest import DeviceIngest
cp, tables, evs0, _ = rig.make_esl_like(row_stride=13)
stream, _ = rig.render_stream(cp, tables, n_frames=16, row_stride=13, seed=9)
with XMapsEngine(tables) as eng:
    pin = eng.host_empty((len(stream),), S.EVENT_CD_DTYPE)
    pin[:] = stream
    packet = int(1e6 / 60 / 4)
    cuts = np.searchsorted(pin["t"], np.arange(pin["t"][0], pin["t"][-1] + packet, packet))
    for rep in range(2):
        with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 18, expected_events_per_frame=150_000, result_ring=32,
                          want_depth=False, want_bgr=True) as ing:
            for a, b in zip(cuts[:4], cuts[1:5]):
                ing.push_pinned(pin[a:b])
            ing.flush(), ing.reset(), ing.poll()
            c0 = time.perf_counter()
            for a, b in zip(cuts[:-1], cuts[1:]):
                ing.push_pinned(pin[a:b])
            ca = time.perf_counter()
            ing.flush()
            c1 = time.perf_counter()
            pr = cProfile.Profile(); pr.enable()
            got = ing.poll(copy=False)
            pr.disable()
            c2 = time.perf_counter()
            print(f"push loop {(ca - c0) * 1e3:.2f} ms, flush {(c1 - ca) * 1e3:.2f} ms, poll {(c2 - c1) * 1e3:.2f} ms, {len(got)} frames")
            pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
