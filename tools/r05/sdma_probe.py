#!/usr/bin/env python3
"""Which SDMA engine the HIP runtime gives the ingest's copies on the process's FIRST set of streams and on a LATER set
(XM_INGEST_OWN_STREAMS), and what the records / EVT 3.0 streams run at on each.  Run with the runtime's copy log on:

    AMD_LOG_LEVEL=4 AMD_LOG_MASK=2304 python tools/r05/sdma_probe.py 2> log.txt      # 256 = LOG_COPY, 2048 = LOG_INIT

The script prints markers to stderr between its phases; tools/r05/sdma_summary.py counts the `copy_engine=` values per phase.
Filter OFF (the comparison is about the copies, and round 4's numbers were taken that way).
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from x_maps_amd import XMapsEngine, rig, evt3, synthetic as S
from x_maps_amd.ingest import DeviceIngest
from x_maps_amd import _native as N


def mark(s):
    sys.stderr.write(f"\n#### PHASE {s}\n")
    sys.stderr.flush()


cp, tables, _, _ = rig.make_esl_like(row_stride=13)
stream, _ = rig.render_stream(cp, tables, n_frames=48, row_stride=13, seed=9)
order = sys.argv[1] if len(sys.argv) > 1 else "first,later,later2"
with XMapsEngine(tables) as eng:
    pin = eng.host_empty((len(stream),), S.EVENT_CD_DTYPE)
    pin[:] = stream
    q = int(1e6 / 60 / 4)
    c4 = np.searchsorted(pin["t"], np.arange(pin["t"][0], pin["t"][-1] + q, q))
    c1 = np.searchsorted(pin["t"], np.arange(pin["t"][0], pin["t"][-1] + 4 * q, 4 * q))
    chunks = []
    for a, b in zip(c1[:-1], c1[1:]):
        if b > a:
            w = evt3.encode_evt3_singles(stream[a:b])
            pw = eng.host_empty(w.shape, np.uint16)
            pw[:] = w
            chunks.append(pw)
    for which in order.split(","):
        N.debug_option("XM_INGEST_OWN_STREAMS", None if which == "first" else "1")
        N.debug_option("XM_INGEST_TRACE", "1" if os.environ.get("TRACE") else None)
        N.debug_option("XM_INGEST_HOST_SEQ", os.environ.get("HOST_SEQ"))
        N.debug_option("XM_INGEST_EVT3_OUT_STREAM", "1")  # (EVT 3.0 frames' copies on the out stream too: the case that differed)
        with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 19, result_ring=64, want_depth=False, activity_filter=False) as ing, \
                evt3.DeviceEvt3Decoder(eng, max_words=max(len(c) for c in chunks)) as dec:
            rec, ev3 = [], []
            for rep in range(5):
                ing.reset(), ing.poll(copy=False)
                if rep == 4:
                    mark(f"{which} records")
                c0 = time.perf_counter()
                for a, b in zip(c4[:-1], c4[1:]):
                    ing.push_pinned(pin[a:b])
                ing.flush()
                ing.poll(copy=False)
                rec.append(len(stream) / (time.perf_counter() - c0) / 1e6)
            for rep in range(5):
                ing.reset(), ing.poll(copy=False), dec.reset()
                if rep == 4:
                    mark(f"{which} evt3")
                c0 = time.perf_counter()
                for c in chunks:
                    dec.push(ing, c, pinned=True, count=False)
                ing.flush()
                ing.poll(copy=False)
                ev3.append(len(stream) / (time.perf_counter() - c0) / 1e6)
            mark("idle")
        print(f"{which:7s} set of streams: records {np.median(rec[1:]):7.1f} Mev/s ({' '.join(f'{v:.0f}' for v in rec)})   "
              f"EVT 3.0 period chunks {np.median(ev3[1:]):7.1f} Mev/s ({' '.join(f'{v:.0f}' for v in ev3)})", flush=True)
