#!/usr/bin/env python3
"""Soak of the device EVT 3.0 decoder against the host decoder: random encoded streams and uniformly random words, whole and in
random chunks, over many seeds:  python tools/evt3_soak.py [first_seed=0] [n_seeds=200]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from x_maps_amd import XMapsEngine, evt3, synthetic as S
import test_gpu_evt3 as T
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
t0 = time.time()
with XMapsEngine(S.make_tables(S.C_TINY)) as eng, evt3.DeviceEvt3Decoder(eng, max_words=1 << 19, max_events=1 << 21) as dec:
    for seed in range(first, first + n):
        rng = np.random.default_rng(31_000 + seed)
        if seed % 2:
            words = rng.integers(0, 65536, int(rng.integers(1, 200_000))).astype("<u2")
            sel = rng.random(len(words)) < rng.random()
            words[sel] = ((rng.choice([0x0, 0x2, 0x3, 0x4, 0x5, 0x6, 0x8], int(sel.sum())) << 12) | rng.integers(0, 4096, int(sel.sum()))).astype("<u2")
        else:
            words = T._random_stream(seed, int(rng.integers(2_000, 40_000)))
        ref = evt3.decode_evt3(words)
        def report(tag, g, h, w):
            print("MISMATCH seed", seed, tag, "events", len(g), len(h), "words", len(w), flush=True)
            if len(g) == len(h):
                for k in "xypt":
                    d = np.nonzero(g[k] != h[k])[0]
                    if len(d):
                        print("   ", k, len(d), "first at", d[:4], "got", g[k][d[:4]], "want", h[k][d[:4]], flush=True)
        dec.reset()
        g = dec.decode(words)
        ok = T._same(g, ref)
        if not ok:
            report("whole", g, ref, words)
        dec.reset()
        host = evt3.Evt3Decoder()
        cuts = np.unique(np.concatenate(([0, len(words)], rng.integers(0, len(words) + 1, int(rng.integers(1, 20))))))
        for a, b in zip(cuts[:-1], cuts[1:]):
            g, h = dec.decode(words[a:b]), host.decode(words[a:b])
            if not T._same(g, h):
                report(f"chunk [{a}, {b}) of cuts {cuts.tolist()}", g, h, words[a:b])
                print("    first words of the chunk:", [hex(int(x)) for x in words[a:a + 12]], "host state after:", host.y, host.base_x, host.base_p, host.t_high, host.t_low, host.t_loops, flush=True)
                ok = False
                break
        if not ok:
            bad += 1
print(f"{n} seeds in {time.time() - t0:.1f} s, mismatches: {bad}")
sys.exit(1 if bad else 0)
