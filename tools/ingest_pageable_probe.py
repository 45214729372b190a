#!/usr/bin/env python3
"""ESL-like stream as PAGEABLE packets (what Metavision hands over) through DeviceIngest.push: what the calling thread pays per
packet (the copy into the pinned staging entry) and the end-to-end rate, beside push_pinned."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from x_maps_amd import XMapsEngine, rig, synthetic as S
from x_maps_amd.ingest import DeviceIngest
if os.environ.get("PROBE_TRACE") == "1":
    from x_maps_amd import _native as _N
    _N.debug_option("XM_INGEST_TRACE", "1")
cp, tables, _, _ = rig.make_esl_like(row_stride=13)
stream, _ = rig.render_stream(cp, tables, n_frames=48, row_stride=13, seed=9)
with XMapsEngine(tables) as eng:
    pin = eng.host_empty((len(stream),), S.EVENT_CD_DTYPE)
    pin[:] = stream
    q = int(1e6 / 60 / 4)
    cuts = np.searchsorted(stream["t"], np.arange(stream["t"][0], stream["t"][-1] + q, q))
    pageable = [np.array(pin[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]  # (16-byte EventCD records, as Metavision hands them over)
    pinned = [pin[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    for label, pk, fn in (("pinned  ", pinned, "push_pinned"), ("pageable", pageable, "push")):
        with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 18, result_ring=64, want_depth=False) as ing:
            res = []
            for rep in range(5):
                ing.reset(), ing.poll(copy=False)
                hs0 = ing.host_stats()
                c0 = time.perf_counter()
                f = getattr(ing, fn)
                for p in pk:
                    f(p)
                c1 = time.perf_counter()
                ing.flush()
                got = ing.poll(copy=False)
                dt = time.perf_counter() - c0
                hs = ing.host_stats()
                res.append((len(stream) / dt / 1e6, (c1 - c0) / len(pk) * 1e6,
                            (hs["host_seconds_in_push"] - hs0["host_seconds_in_push"] - hs["seconds_waiting_for_the_gpu"] + hs0["seconds_waiting_for_the_gpu"]) / len(pk) * 1e6))
            res.sort()
            print(f"{label}: {res[2][0]:.0f} Mev/s (median of 5), push loop {res[2][1]:.1f} us per packet, library time in push excluding back-pressure {res[2][2]:.1f} us per packet, {len(got)} frames", flush=True)
