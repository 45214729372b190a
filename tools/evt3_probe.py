#!/usr/bin/env python3
"""Decode throughput of the device EVT 3.0 decoder (and, behind it, of the EVT 2.0 one) on a chunk of a camera-like stream
(host -> device words included):
  python tools/evt3_probe.py [events=2000000] [reps=20]      (under rocprofv3 --kernel-trace --stats for the kernels' own times)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from x_maps_amd import XMapsEngine, evt3, synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
evs = S.make_events(S.C_1M, frame=0, n=n)
t0 = time.perf_counter(); words = evt3.encode_evt3_singles(evs); t1 = time.perf_counter()
ref = evt3.Evt3Decoder()
t2 = time.perf_counter(); host = ref.decode(words); t3 = time.perf_counter()
print(f"{n} events = {len(words)} words ({2 * len(words) / n:.2f} B/event); host decoder {n / (t3 - t2) / 1e6:.1f} Mev/s")
with XMapsEngine(S.make_tables(S.C_TINY)) as eng, evt3.DeviceEvt3Decoder(eng, max_words=len(words), max_events=n + 64) as dec:
    got = dec.decode(words)
    assert all(np.array_equal(got[k], host[k]) for k in ("x", "y", "p", "t"))
    dec.reset(); dec.decode_device(words)
    c0 = time.perf_counter()
    for _ in range(reps):
        dec.decode_device(words)
    dt = (time.perf_counter() - c0) / reps
    print(f"device decoder: {dt * 1e3:.3f} ms per chunk = {n / dt / 1e6:.0f} Mev/s, {2 * len(words) / dt / 1e9:.1f} GB/s of words (pageable host memory -> records in HBM)")

from x_maps_amd import evt2
words2 = evt2.encode_evt2(evs, time_high_every_us=16)
t2 = time.perf_counter(); host2 = evt2.Evt2Decoder().decode(words2); t3 = time.perf_counter()
print(f"EVT 2.0: {n} events = {len(words2)} words ({4 * len(words2) / n:.2f} B/event); host decoder {n / (t3 - t2) / 1e6:.1f} Mev/s")
with XMapsEngine(S.make_tables(S.C_TINY)) as eng, evt2.DeviceEvt2Decoder(eng, max_words=len(words2), max_events=n + 64) as dec:
    got = dec.decode(words2)
    assert all(np.array_equal(got[k], host2[k]) for k in ("x", "y", "p", "t"))
    dec.reset(); dec.decode_device(words2)
    c0 = time.perf_counter()
    for _ in range(reps):
        dec.decode_device(words2)
    dt = (time.perf_counter() - c0) / reps
    print(f"EVT 2.0 device decoder: {dt * 1e3:.3f} ms per chunk = {n / dt / 1e6:.0f} Mev/s, {4 * len(words2) / dt / 1e9:.1f} GB/s of words (pageable host memory -> records in HBM)")
