#!/usr/bin/env python3
"""Soak of the device ingest (event ring, verdicts, launch thread, frame stream + DMA): random streams in random
packets through random configurations -- launch thread on / off, activity filter on (thresholds from one period down to 0.7 ms:
one to many time buckets per packet, the sequential path; packets whose stamps step back) / off, rings from tiny (many wrap-arounds,
no run-ahead) to roomy (run-ahead 3), pageable / pinned packets, EVT 3.0 / EVT 2.0 words with the count left on the device, polling after
every push or only at the end -- against the CPU chain (oracle/ingest_oracle.py + xmaps_oracle.py): the same frames (first / last
stamp, length, inliers, depth) every time.

    python tools/ingest_soak.py [first_seed=0] [n=200]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import ingest_oracle as IO
import xmaps_oracle as O
from x_maps_amd import XMapsEngine, evt2, evt3, synthetic as S
from x_maps_amd.ingest import DeviceIngest
import test_gpu_ingest as TI

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = S.C_TINY
tb = S.make_tables(cfg)
bad = []
t0 = time.time()
with XMapsEngine(tb) as eng:
    for seed in range(first, first + n_runs):
        rng = np.random.default_rng(10_000 + seed)
        n_frames = int(rng.integers(6, 20))
        stream = TI._tiny_stream(n_frames, seed=seed, per_frame=int(rng.integers(1500, 4000)), neg=float(rng.choice([0.0, 0.1, 0.4])),
                                 gap_noise=int(rng.choice([0, 2, 3])))
        mode = rng.choice(["time", "count"])
        if mode == "time":
            pk = TI._packets(stream, int(rng.choice([1000, 2000, 4166, 8000, 16_600])))
        else:
            cuts = [0]
            while cuts[-1] < len(stream):
                cuts.append(min(len(stream), cuts[-1] + int(rng.choice([1, 3, 64, 700, 2048, 2049, 5000]))))
            pk = [stream[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
        if rng.random() < 0.3:
            pk = [p for q in pk for p in (q, stream[:0])][:len(pk) + 5]  # empty packets in between
        activity = bool(rng.random() < 0.5)
        act_thresh = int(rng.choice([int(1e6 / 60), int(1e6 / 60), 3000, 700])) if activity else 0  # (700 us: > 8 buckets in many packets)
        thread = bool(rng.random() < 0.7)
        words = bool(rng.random() < 0.35)
        if activity and not words and mode == "time" and rng.random() < 0.3 and len(pk) > 4:
            # a glitch: one packet's second half steps back in time (the device judges that packet sequentially)
            k = int(rng.integers(1, len(pk) - 1))
            if len(pk[k]) > 8:
                gl = pk[k].copy()
                gl["t"][len(gl) // 2:] -= int(rng.choice([50, 5_000, 40_000]))
                pk[k] = gl
        fmt = 2 if (words and rng.random() < 0.5) else 3  # EVT 2.0 or EVT 3.0 words
        pinned = bool(rng.random() < 0.5)
        poll_each = bool(rng.random() < 0.5)
        max_pk = max(2048, 1 << int(np.ceil(np.log2(max(len(p) for p in pk) + 1))))
        cap = max_pk * int(rng.choice([2, 4, 8, 16, 32]))
        tf = IO.TriggerFinderOracle(60)
        act = IO.ActivityFilterC(cfg.cam_w, cfg.cam_h, act_thresh or int(1e6 / 60))
        for p in pk:
            pos = IO.polarity_filter(p)
            tf.process_events(act.process(pos) if activity else pos)
        want = tf.frames
        # (a ring that cannot hold what the trigger finder may keep drops and says so: those runs only check that it says so)
        got = []
        keep = []
        with DeviceIngest(eng, 60, activity_filter=activity, activity_thresh_us=act_thresh, capacity_events=cap, max_packet_events=max_pk,
                          result_ring=64, launch_thread=thread) as ing:
            dec = (evt2.DeviceEvt2Decoder(eng, max_words=8 * max_pk) if fmt == 2 else evt3.DeviceEvt3Decoder(eng, max_words=8 * max_pk)) if words else None
            for p in pk:
                if words:
                    w = evt2.encode_evt2(p, time_high_every_us=16) if fmt == 2 else evt3.encode_evt3(p)
                    if pinned and len(w):
                        pw = eng.host_empty(w.shape, w.dtype)
                        pw[:] = w
                        keep.append(pw)
                        dec.push(ing, pw, pinned=True, count=False)
                    else:
                        dec.push(ing, w, count=bool(rng.random() < 0.5))
                elif pinned and len(p):
                    pp = eng.host_empty((len(p),), S.EVENT_CD_DTYPE)
                    pp[:] = p
                    keep.append(pp)
                    ing.push_pinned(pp)
                else:
                    ing.push(p)
                if poll_each:
                    got += ing.poll()
            ing.flush()
            got += ing.poll()
            dstat = ing.device_stats()
            if dec is not None:
                dec.close()
        desc = dict(seed=seed, frames=n_frames, packets=len(pk), mode=mode, activity=activity, act_thresh=act_thresh, thread=thread, words=words, fmt=fmt, pinned=pinned,
                    poll_each=poll_each, cap=cap, max_pk=max_pk)
        overflow = max([f.overflow for f in got] + [dstat["events_dropped"]])
        if overflow or any(f.lost for f in got):
            if cap >= 16 * max_pk and not any(f.lost for f in got):
                bad.append(("overflow on a roomy ring", desc))
            continue
        ok = len(got) == len(want)
        for f, evs in zip(got, want):
            if not ok:
                break
            ok = (f.n_events, f.t_first, f.t_last) == (len(evs), int(evs["t"][0]), int(evs["t"][-1]))
            if ok:
                x, y, t, _ = S.to_soa(evs)
                ref = O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)
                ok = f.n_inliers == int(ref["mask"].sum()) and np.array_equal(f.depth, ref["depth"]) and np.array_equal(f.bgr, ref["bgr"])
        if not ok:
            bad.append(("frames differ", desc, len(got), len(want)))
            print("MISMATCH", desc, len(got), len(want), flush=True)
        for b in keep:
            pass
print(f"{n_runs} runs in {time.time() - t0:.1f} s: {len(bad)} bad")
for b in bad:
    print(b)
sys.exit(1 if bad else 0)
