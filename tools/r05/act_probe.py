#!/usr/bin/env python3
"""The ESL-like camera stream through the device ingest with the activity filter on / off (records, quarter-period packets):
rate, frames, and -- under rocprofv3 --kernel-trace --stats -- what each ingest kernel costs per packet.

    python tools/r05/act_probe.py [filter=1] [passes=3]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from x_maps_amd import XMapsEngine, rig, synthetic as S
from x_maps_amd.ingest import DeviceIngest
act = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
from x_maps_amd import _native as N
if os.environ.get("NO_COPY_THREAD"):
    N.debug_option("XM_INGEST_NO_COPY_THREAD", "1")
if os.environ.get("TRACE"):
    N.debug_option("XM_INGEST_TRACE", "1")
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cp, tables, _, _ = rig.make_esl_like(row_stride=13)
stream, _ = rig.render_stream(cp, tables, n_frames=48, row_stride=13, seed=9)
with XMapsEngine(tables) as eng:
    pin = eng.host_empty((len(stream),), S.EVENT_CD_DTYPE)
    pin[:] = stream
    packet = int(1e6 / 60 / 4)
    cuts = np.searchsorted(pin["t"], np.arange(pin["t"][0], pin["t"][-1] + packet, packet))
    with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 18, expected_events_per_frame=150_000, result_ring=64,
                      want_depth=False, want_bgr=True, activity_filter=act) as ing:
        for rep in range(passes + 1):
            c0 = time.perf_counter()
            for a, b in zip(cuts[:-1], cuts[1:]):
                ing.push_pinned(pin[a:b])
            ing.flush()
            got = ing.poll(copy=False)
            dt = time.perf_counter() - c0
            st = ing.device_stats()
            ing.reset()
            print(f"filter={act} pass {rep}: {len(got)} frames, {len(stream) / dt / 1e6:.1f} Mev/s, {dt / max(len(got), 1) * 1e3:.3f} ms per frame, "
                  f"{(len(cuts) - 1)} packets = {dt / (len(cuts) - 1) * 1e6:.1f} us per packet, appended {st['events_appended']}")
