#!/usr/bin/env python3
"""ESL-like stream as EVT 3.0 words in period chunks (one frame per chunk) through the decoder + device ingest: rate for a few sizes
of the result frames' D2H pieces (XM_INGEST_OUT_PIECE).   python tools/esl_evt3_probe.py [piece_bytes ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_TORCH") == "1":
    import torch
    torch.cuda.init()
import numpy as np
from x_maps_amd import XMapsEngine, rig, evt3, synthetic as S
from x_maps_amd.ingest import DeviceIngest
from x_maps_amd import _native as N
cp, tables, _, _ = rig.make_esl_like(row_stride=13)
stream, _ = rig.render_stream(cp, tables, n_frames=48, row_stride=13, seed=9)
if os.environ.get("PROBE_TRACE") == "1":
    N.debug_option("XM_INGEST_TRACE", "1")
pieces = [int(a) for a in sys.argv[1:]] or [4 << 20]  # (a negative size: the copies on the frame stream, XM_INGEST_OUT_SERIAL, whole)
with XMapsEngine(tables) as eng:
    packet = int(1e6 / 60)
    cuts = np.searchsorted(stream["t"], np.arange(stream["t"][0], stream["t"][-1] + packet, packet))
    chunks = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            w = evt3.encode_evt3_singles(stream[a:b])
            pw = eng.host_empty(w.shape, np.uint16)
            pw[:] = w
            chunks.append(pw)
    pin = eng.host_empty((len(stream),), S.EVENT_CD_DTYPE)
    pin[:] = stream
    q = int(1e6 / 60 / 4)
    c4 = np.searchsorted(pin["t"], np.arange(pin["t"][0], pin["t"][-1] + q, q))
    for piece in pieces:
        N.debug_option("XM_INGEST_OUT_PIECE", str(abs(piece)))
        N.debug_option("XM_INGEST_OUT_SERIAL", "1" if piece < 0 else "0")
        N.debug_option("XM_INGEST_PRIOS", os.environ.get("PROBE_PRIOS"))
        N.debug_option("XM_INGEST_OUT_INLINE", "1" if os.environ.get("PROBE_OUT_INLINE") == "1" else None)
        if os.environ.get("PROBE_RECORDS_FIRST") == "1":
            with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 18, result_ring=64, want_depth=False) as ing:
                rec = []
                for rep in range(7):
                    ing.reset(), ing.poll(copy=False)
                    c0 = time.perf_counter()
                    for a, b in zip(c4[:-1], c4[1:]):
                        ing.push_pinned(pin[a:b])
                    ing.flush()
                    g2 = ing.poll(copy=False)
                    rec.append(len(stream) / (time.perf_counter() - c0) / 1e6)
            with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 19, result_ring=64, want_depth=False) as ing, \
                    evt3.DeviceEvt3Decoder(eng, max_words=max(len(c) for c in chunks)) as dec:
                res = []
                for rep in range(7):
                    ing.reset(), ing.poll(copy=False), dec.reset()
                    c0 = time.perf_counter()
                    for c in chunks:
                        dec.push(ing, c, pinned=True, count=False)
                    ing.flush()
                    got = ing.poll(copy=False)
                    res.append(len(stream) / (time.perf_counter() - c0) / 1e6)
        else:
            with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 19, result_ring=64, want_depth=False) as ing, \
                    evt3.DeviceEvt3Decoder(eng, max_words=max(len(c) for c in chunks)) as dec:
                res = []
                for rep in range(7):
                    ing.reset(), ing.poll(copy=False), dec.reset()
                    c0 = time.perf_counter()
                    for c in chunks:
                        dec.push(ing, c, pinned=True, count=False)
                    ing.flush()
                    got = ing.poll(copy=False)
                    res.append(len(stream) / (time.perf_counter() - c0) / 1e6)
            with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 18, result_ring=64, want_depth=False) as ing:
                rec = []
                for rep in range(7):
                    ing.reset(), ing.poll(copy=False)
                    c0 = time.perf_counter()
                    for a, b in zip(c4[:-1], c4[1:]):
                        ing.push_pinned(pin[a:b])
                    ing.flush()
                    g2 = ing.poll(copy=False)
                    rec.append(len(stream) / (time.perf_counter() - c0) / 1e6)
        med = lambda v: f"median {np.median(v[2:]):.0f} ({min(v[2:]):.0f}-{max(v[2:]):.0f}; first passes {v[0]:.0f}, {v[1]:.0f})"
        print(f"piece {piece}: evt3 period chunks, {len(got)} frames: {med(res)} Mev/s | records, quarter-period packets, {len(g2)} frames: {med(rec)} Mev/s", flush=True)
