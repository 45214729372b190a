"""bench.py --graph: BASELINE configs[4], 60 x C-1M frames replayed from one captured hipGraph."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

from .common import (BENCH_PY, HBM_PEAK_GBS, PREWARM_S, ROOT, TARGET_TIMED_S, Timer, apply_measured_traffic, cpu_baseline_leg, depth_parity,
                     n_blocks_for, parity_ok, pipeline_fractions, roofline_dict, roofline_of, traffic_file_age)


# =====================================================================================================================
# --graph: configs[4], 60 frames x 1 M events captured once into a hipGraph and replayed
# =====================================================================================================================
def bench_graph(args, torch, dist, dev, rank, local_rank, world):
    from x_maps_amd import XMapsEngine
    from x_maps_amd import synthetic as S

    cfg = S.C_1M
    tables = S.make_tables(cfg)
    camera = args.camera_perspective
    F = 60
    slots = args.slots or F
    n_ev = cfg.n_events
    eng = XMapsEngine(tables, camera_perspective=camera, device=local_rank, n_slots=slots, default_priority_streams=True,
                      assume_time_sorted=args.assume_sorted)
    H, W = eng.out_h, eng.out_w
    X = torch.empty(F * n_ev, dtype=torch.int16, device=dev)
    Y = torch.empty_like(X)
    T = torch.empty(F * n_ev, dtype=torch.int64, device=dev)
    host = {}
    for f in range(F):  # seeds 20230 .. 20289; other ranks take the next 60
        x, y, t, _ = S.to_soa(S.make_events(cfg, frame=rank * F + f))
        if f in (0, F - 1):
            host[f] = (x, y, t)
        X[f * n_ev:(f + 1) * n_ev] = torch.from_numpy(x.view(np.int16))
        Y[f * n_ev:(f + 1) * n_ev] = torch.from_numpy(y.view(np.int16))
        T[f * n_ev:(f + 1) * n_ev] = torch.from_numpy(t)
    depth = torch.zeros((F, H, W), dtype=torch.float32, device=dev)
    bgr = None if args.no_bgr else torch.zeros((F, H, W, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    offs = np.arange(F + 1, dtype=np.uint64) * n_ev
    graph = eng.graph_create(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(),
                             None if bgr is None else bgr.data_ptr())
    paths = eng.path_counts()  # which K1 the frames were captured with
    one = eng.graph_create(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs[:2], depth.data_ptr(),
                           None if bgr is None else bgr.data_ptr())
    parity = None
    O = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import xmaps_oracle as O
        graph.launch()
        eng.sync()
        parity = {}
        for f, (x, y, t) in host.items():
            ref = O.process_ev_frame(tables, x.astype(np.int64), y.astype(np.int64), t, camera_perspective=camera,
                                     want_bgr=bgr is not None)
            pf = depth_parity(depth[f].cpu().numpy(), ref["depth"])
            if bgr is not None:
                pf["bgr_equal"] = bool(np.array_equal(bgr[f].cpu().numpy(), ref["bgr"]))
            parity[f"frame_{f}"] = pf
            if not (pf["depth_max_rel_err"] <= 1e-4 and pf["empty_mask_equal"] and pf.get("bgr_equal", True)) and not args.no_parity:
                print(json.dumps({"error": "parity check failed", "parity": parity}))
                sys.exit(1)
    tm = Timer(torch, dist, dev, eng.sync)
    replays = max(1, (args.steps + F - 1) // F)
    steps = replays * F
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < PREWARM_S:
        graph.launch()
        eng.sync()
    # latency: one replay at a time, synchronised (host clock around launch .. sync)
    lat = []
    for _ in range(200):
        c0 = time.perf_counter()
        graph.launch()
        eng.sync()
        lat.append(time.perf_counter() - c0)
    lat1 = []
    for _ in range(1000):
        c0 = time.perf_counter()
        one.launch()
        eng.sync()
        lat1.append(time.perf_counter() - c0)
    lat, lat1 = np.array(lat) * 1e6, np.array(lat1) * 1e6
    est = tm.agree(float(np.median(lat)) * 1e-6 / F)
    R = int(min(100, max(3, round(TARGET_TIMED_S / max(steps * est, 1e-6))))) if not args.single_block else 1
    el, enq = tm.blocks(lambda: [graph.launch() for _ in range(replays)], R)
    elapsed = float(np.median(el))
    value = float(n_ev) * steps * world / elapsed / 1e6
    if rank != 0:
        graph.close(), one.close(), eng.close()
        return None
    # roofline: the graph's kernel nodes cannot carry events of their own, so the same three multi-frame kernels (boundary pass,
    # K1, K2: the same grids over the same 60 frames and slots) are launched eagerly with HIP events attached to each dispatch
    roofline = None
    if slots >= F:
        def prof_group(i):
            return eng.profile_batch_device(X.data_ptr(), Y.data_ptr(), T.data_ptr(), None, offs, depth.data_ptr(),
                                            None if bgr is None else bgr.data_ptr())
        roofline, alg, pt, wl = roofline_of(eng, None, n_ev, (None, None), tables, camera, 0 if bgr is None else 3, world,
                                            (F, prof_group), cell_bytes=2 if paths["cols"] else (4 if paths["key32"] else 8))
        roofline["timing"] += ("; --graph: these are the graph's first three kernel nodes launched eagerly (the captured batch adds "
                               "the four redo nodes, which return at once for frames whose tiles held)")
        pipeline_fractions(roofline, alg, pt, wl, value, world, elapsed / steps, F, helper_runs=True)
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline_leg(args, O, tables, host[0], n_ev, camera, bgr is not None)
    out = {
        "metric": "Mevents/s to depth frame, 640x480, 1M ev/frame", "value": round(value, 2), "unit": "Mevents/s",
        "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
        "config": {"workload": "C-60x1M: 60 frames x 1M events (seeds 20230..20289), 640x480 cam/proj, one captured hipGraph, "
                               "1xMI355X" + (" (camera view)" if camera else " (projector view)"),
                   "events_per_frame": n_ev, "frames_per_graph": F, "key_frames": slots,
                   "graph_nodes": ("7 multi-frame kernel nodes, grid = 60 frames x tiles: K0b, K1 column tiles, K2 on the u16 frame (frames "
                                   "whose tiles held) + counters reset, K0, K1, K2 on the 64-bit key frame (frames whose tiles objected; "
                                   "every other block returns at once)" if paths["cols"] else
                                   "3 multi-frame kernel nodes (K0, K1, K2: grid = 60 frames x tiles)") if slots >= F else
                                  f"groups of {slots // 2} frames, alternating between two graph branches",
                   "k1_paths_frames_captured": paths,
                   "extrema": "XM_FLAG_TIME_SORTED (no K0)" if args.assume_sorted else
                              "column tiles with the redo decided on the device (no host at hand inside a graph); XM_COLS=0: extrema pass K0 + 64-bit keys",
                   "steps_note": f"a step = one frame; --steps rounded up to {replays} replay(s) of the 60-frame graph",
                   "launch": "hipGraph"},
        "latency_us": {"batch_of_60_frames": {"p50": round(float(np.percentile(lat, 50)), 1), "p99": round(float(np.percentile(lat, 99)), 1),
                                              "per_frame_amortised_p50": round(float(np.percentile(lat, 50)) / F, 2)},
                       "single_frame_graph": {"p50": round(float(np.percentile(lat1, 50)), 1), "p99": round(float(np.percentile(lat1, 99)), 1)},
                       "definition": "host clock from xm_graph_launch to the return of xm_sync (events resident in HBM -> depth+BGR "
                                     "resident in HBM), one replay at a time; 200 replays of the 60-frame graph, 1000 of a 1-frame graph"},
        "timing": {"prewarm_s": PREWARM_S, "blocks": int(R), "block_s_median": round(elapsed, 6),
                   "block_s_min": round(float(el.min()), 6), "block_s_max": round(float(el.max()), 6)},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
    }
    graph.close(), one.close(), eng.close()
    return out
