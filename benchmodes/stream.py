"""bench.py, default mode: BASELINE configs[1], C-1M frames in groups of 32 (+ the host / ingest legs of the default line)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

from .other import other_config_legs
from .common import leg_clock
from .common import (BENCH_PY, HBM_PEAK_GBS, PREWARM_S, ROOT, TARGET_TIMED_S, Timer, apply_measured_traffic, cpu_baseline_leg, depth_parity,
                     n_blocks_for, parity_ok, pipeline_fractions, roofline_dict, roofline_of, traffic_file_age)


# =====================================================================================================================
# default: configs[1], C-1M frames streamed through the asynchronous device-pointer path
# =====================================================================================================================
def bench_stream(args, torch, dist, dev, rank, local_rank, world):
    from x_maps_amd import XMapsEngine
    from x_maps_amd import synthetic as S

    cfg = S.C_1M
    tables = S.make_tables(cfg)
    camera = args.camera_perspective
    B = args.batch
    adaptive = not B and not args.no_adaptive  # one frame per call: frames that arrive while the GPU is busy go out as groups
    slots = args.slots or (max(4, args.groups_in_flight * B) if B else (64 if adaptive else 4))
    mode_kw = {"force_general": args.general, "assume_time_sorted": args.assume_sorted}
    # groups are launched by the calling thread (three launches per GROUP); one frame per call without adaptive batching uses the
    # launch workers
    eng = XMapsEngine(tables, camera_perspective=camera, device=local_rank, n_slots=slots, adaptive_batch=adaptive,
                      launch_workers=(not args.no_launch_workers) and not B and not adaptive, **mode_kw)
    H, W = eng.out_h, eng.out_w
    n_ev = cfg.n_events

    leg_clock('start')
    # ---- synthetic frames -> HBM (SoA columns, the layout K1 reads), laid out back to back ------------------------
    nf = args.frames or (args.groups_in_flight * B if B else 32)
    if B and (nf % B or slots % B):
        raise SystemExit("--batch must divide --frames and the number of slots")
    X = torch.empty(nf * n_ev, dtype=torch.int16, device=dev)
    Y = torch.empty_like(X)
    T = torch.empty(nf * n_ev, dtype=torch.int64, device=dev)
    host_frames = {}
    keep = {0, 1, 2, 3, max(B - 1, 0)}  # frames kept on the host for the parity checks
    for f in range(nf):
        evs = S.make_events(cfg, frame=rank * nf + f)
        x, y, t, _ = S.to_soa(evs)
        if f in keep:
            host_frames[f] = (x, y, t)
        X[f * n_ev:(f + 1) * n_ev] = torch.from_numpy(x.view(np.int16))
        Y[f * n_ev:(f + 1) * n_ev] = torch.from_numpy(y.view(np.int16))
        T[f * n_ev:(f + 1) * n_ev] = torch.from_numpy(t)
    frames = [(X[f * n_ev:], Y[f * n_ev:], T[f * n_ev:]) for f in range(nf)]
    n_out = max(slots, 1)
    depth_out = torch.empty((n_out, H, W), dtype=torch.float32, device=dev)
    bgr_out = None if args.no_bgr else torch.empty((n_out, H, W, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    bgr_b = 0 if bgr_out is None else 3
    key_mb = eng.key_shape[0] * eng.key_shape[1] * (8 + 4 + 2) / 1e6  # 64-bit + compact key frame + u16 disparity frame per slot
    resident_mb = nf * n_ev * 12 / 1e6 + slots * key_mb + n_out * H * W * (4 + bgr_b) / 1e6
    fps = B or 1  # frames per step

    def oracle_frame(f, cam, want_bgr):
        hx, hy, ht = host_frames[f]
        return O.process_ev_frame(tables, hx.astype(np.int64), hy.astype(np.int64), ht, camera_perspective=cam, want_bgr=want_bgr)

    def make_step(e, d_out, b_out, nsl, Bm):
        """step(i): the i-th step = group i of Bm consecutive resident frames (Bm > 0) or frame i (Bm == 0)."""
        if Bm:
            offs = np.arange(Bm + 1, dtype=np.uint64) * n_ev
            gptr = [(X[g * Bm * n_ev:].data_ptr(), Y[g * Bm * n_ev:].data_ptr(), T[g * Bm * n_ev:].data_ptr()) for g in range(nf // Bm)]
            optr = [(d_out[o * Bm].data_ptr(), None if b_out is None else b_out[o * Bm].data_ptr()) for o in range(nsl // Bm)]
            call = e.process_batch_device

            def step_group(i):
                gx, gy, gt = gptr[i % len(gptr)]
                d, b = optr[i % len(optr)]
                call(gx, gy, gt, None, offs, d, b)
            return step_group

        # raw device pointers are taken once (a host holds them anyway); the step itself is one C-ABI call
        fptr = [(fx.data_ptr(), fy.data_ptr(), ft.data_ptr()) for fx, fy, ft in frames]
        optr = [(d_out[o].data_ptr(), None if b_out is None else b_out[o].data_ptr()) for o in range(nsl)]
        call = e.process_frame_device

        def step(i):
            fx, fy, ft = fptr[i % nf]
            d, b = optr[i % nsl]
            call(fx, fy, ft, None, n_ev, d, b)
        return step

    def run_steps(step, k, start=0):
        for i in range(start, start + k):
            step(i)

    step = make_step(eng, depth_out, bgr_out, n_out, B)

    leg_clock('frames rendered + uploaded')
    # ---- parity gate before any timing: first (and last) frame of step 0 against the CPU oracle (rank 0) ----------------
    parity = None
    O = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import xmaps_oracle as O  # checker + cpu_baseline only
        step(0)
        eng.sync()
        ref = oracle_frame(0, camera, bgr_out is not None)
        parity = depth_parity(depth_out[0].cpu().numpy(), ref["depth"])
        if bgr_out is not None:
            parity["bgr_equal"] = bool(np.array_equal(bgr_out[0].cpu().numpy(), ref["bgr"]))
        if B > 1:
            refl = oracle_frame(B - 1, camera, False)
            parity["last_frame_of_the_group_depth_bit_exact"] = bool(np.array_equal(depth_out[B - 1].cpu().numpy(), refl["depth"]))
        st = eng.last_frame_stats()  # (the group's last frame)
        parity["n_inliers_equal"] = bool(st.n_inliers == int((refl if B > 1 else ref)["mask"].sum()))
        ok = (parity["depth_max_rel_err"] <= 1e-4 and parity["empty_mask_equal"] and parity["n_inliers_equal"]
              and parity.get("bgr_equal", True) and parity.get("last_frame_of_the_group_depth_bit_exact", True))
        if not ok and args.no_parity:
            parity["IGNORED"] = True
        elif not ok:
            print(json.dumps({"error": "parity check failed", "parity": parity}))
            sys.exit(1)

    tm = Timer(torch, dist, dev, eng.sync)
    leg_clock('parity gate')
    # ---- W warm-up steps, fixed pre-warm, per-kernel profile pass, short re-warm, R timed blocks of exactly K steps ----
    run_steps(step, args.warmup)
    est = tm.prewarm(step, PREWARM_S)
    roofline = alg = pt = wl = None
    if rank == 0:
        group = None
        if B:
            offs_p = np.arange(B + 1, dtype=np.uint64) * n_ev

            def prof_group(i):
                g = i % (nf // B)
                return eng.profile_batch_device(X[g * B * n_ev:].data_ptr(), Y[g * B * n_ev:].data_ptr(), T[g * B * n_ev:].data_ptr(),
                                                None, offs_p, depth_out[0].data_ptr(), None if bgr_out is None else bgr_out[0].data_ptr())
            group = (B, prof_group)
        roofline, alg, pt, wl = roofline_of(eng, frames, n_ev,
                                            (depth_out[0].data_ptr(), None if bgr_out is None else bgr_out[0].data_ptr()),
                                            tables, camera, bgr_b, world, group)
    est = tm.agree(tm.prewarm(step, 0.1))
    R = n_blocks_for(args, est)
    el, enq = tm.blocks(lambda: run_steps(step, args.steps), R)
    elapsed = float(np.median(el))
    total_events = float(n_ev) * fps * args.steps * world
    value = total_events / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    paths = eng.path_counts()
    # (N > 1: the path's real exchange step -- one 10 M-event frame sharded over the same ranks -- is measured by main() BEHIND this
    #  function, under a wall-clock guard, once rank 0 holds the finished replicas' line: sharded_leg_guarded)
    if rank != 0:
        eng.close()
        return None

    s_frame = elapsed / (args.steps * fps)  # seconds per frame, pipelined
    pipeline_fractions(roofline, alg, pt, wl, value, world, s_frame, fps,
                       helper_runs=bool(paths["general"] or paths["cols"] or args.general))
    frames_redone = eng.sorted_fallbacks()
    # roofline.traffic from counters of THIS run (two short child runs of this very workload under rocprofv3 --pmc, one group at
    # a time) -- the committed profiles/pmc_traffic.json only when that is not possible, and then with its age
    if world == 1 and B and not args.no_pmc and not getattr(args, "as_leg", False) and os.environ.get("XM_BENCH_PMC_CHILD") != "1":
        from benchmodes.pmc import measure_traffic
        os.environ["XM_BENCH_PMC_CHILD"] = "1"
        try:
            flags = ["--steps", "6", "--warmup", "2", "--groups-in-flight", "1", "--batch", str(B), "--no-cpu-baseline", "--no-other-modes",
                     "--no-host-path", "--no-other-configs", "--single-block", "--no-pmc"] + (["--camera-perspective"] if camera else []) + \
                    (["--no-bgr"] if args.no_bgr else [])
            measured, detail = measure_traffic(BENCH_PY, flags)
        finally:
            os.environ.pop("XM_BENCH_PMC_CHILD", None)
        apply_measured_traffic(roofline, measured, detail, s_frame, fps)
    elif roofline.get("traffic_source"):
        roofline["traffic_measured_in_run"] = False
        roofline["traffic_source"] += "; " + traffic_file_age()

    leg_clock('timed blocks + profile pass')
    # ---- CPU baseline (rank 0 at N = 1 only) ---------------------------------------------------------------------------
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline_leg(args, O, tables, host_frames[0], n_ev, camera, bgr_out is not None)

    leg_clock('cpu baseline')
    # ---- the same frames with other engine settings (extra information, never the headline `value`) -----------------------
    eng.close()  # one engine at a time: two engines would share the high-priority hardware queues
    other_modes = None
    # (XM_BENCH_FORCE_SHARDED_LEG: the tests' stand-in of an N > 1 run on a one-GPU box -- as there, no extra single-GPU loops)
    if world == 1 and not args.no_other_modes and os.environ.get("XM_BENCH_FORCE_SHARDED_LEG") != "1":
        other_modes = {}
        modes = []  # (name, engine flags, camera view, frames per call, launch workers)
        if B:
            modes.append(("one_frame_per_call", dict(mode_kw, adaptive_batch=True), camera, 0, False))
            modes.append(("one_frame_per_call_eager", dict(mode_kw), camera, 0, not args.no_launch_workers))
        else:
            modes.append(("groups_of_32_frames_per_call", dict(mode_kw), camera, 32, False))
            if not args.no_launch_workers:
                modes.append(("launches_from_the_calling_thread", dict(mode_kw), camera, 0, False))
        if not args.general:
            # (one frame per call, as rounds 1 and 2 reported it)
            modes.append(("forced_general", {"force_general": True}, camera, 0, not args.no_launch_workers))
        if not camera:
            modes.append(("camera_view", dict(mode_kw), True, B, (not args.no_launch_workers) and not B))
        for name, kw, cam, Bm, workers in modes:
            nsl = max(4, args.groups_in_flight * Bm) if Bm else (64 if kw.get("adaptive_batch") else 4)
            e2 = XMapsEngine(tables, camera_perspective=cam, device=local_rank, n_slots=nsl, launch_workers=workers, **kw)
            H2, W2 = e2.out_h, e2.out_w
            d2 = torch.empty((nsl, H2, W2), dtype=torch.float32, device=dev)
            b2 = None if bgr_out is None else torch.empty((nsl, H2, W2, 3), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            step2 = make_step(e2, d2, b2, nsl, Bm)
            per = Bm or 1
            tm2 = Timer(torch, None, dev, e2.sync)
            est2 = tm2.prewarm(step2, PREWARM_S)
            k2 = max(1, args.steps * fps // per)  # the same number of frames as a timed block of the headline
            R2 = int(min(200, max(3, round(0.2 / max(k2 * est2, 1e-6)))))
            el2, _ = tm2.blocks(lambda: run_steps(step2, k2), R2)
            dt = float(np.median(el2))
            # the last step's last frame against the oracle (when that frame is one of those kept on the host)
            li = k2 - 1
            fi = (li % (nf // per)) * per + per - 1
            oi = (li % (nsl // per)) * per + per - 1
            same = None
            if fi in host_frames:
                same = bool(np.array_equal(d2[oi].cpu().numpy(), oracle_frame(fi, cam, False)["depth"]))
            other_modes[name] = {"value": round(n_ev * k2 * per / dt / 1e6, 2), "unit": "Mevents/s",
                                 "ms_per_frame": round(dt / (k2 * per) * 1e3, 5), "blocks": R2, "k1_paths": e2.path_counts(),
                                 "frames_redone_on_general_path": e2.sorted_fallbacks()}
            if same is not None:
                other_modes[name]["depth_equals_oracle"] = same
            e2.close()
        if B and not camera:
            try:
                # the same frames as Metavision hands them over: 16-byte EventCD records (x:u16 y:u16 p:i16 - t:i64), SURVEY 8(a) row A0,
                # consumed unchanged by xm_process_batch_aos (16 instead of 12 bytes per event for K1 to read)
                A = torch.empty((nf * n_ev, 8), dtype=torch.int16, device=dev)
                A[:, 0], A[:, 1], A[:, 2], A[:, 3] = X, Y, 1, 0
                A[:, 4:8] = T.view(torch.int16).reshape(nf * n_ev, 4)
                nsl = max(4, args.groups_in_flight * B)
                e2 = XMapsEngine(tables, camera_perspective=False, device=local_rank, n_slots=nsl, **mode_kw)
                d2 = torch.empty((nsl, H, W), dtype=torch.float32, device=dev)
                b2 = None if bgr_out is None else torch.empty((nsl, H, W, 3), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                offs_a = np.arange(B + 1, dtype=np.uint64) * n_ev
                gptr = [A[g * B * n_ev:].data_ptr() for g in range(nf // B)]
                optr = [(d2[o * B].data_ptr(), None if b2 is None else b2[o * B].data_ptr()) for o in range(nsl // B)]

                def step_aos(i):
                    d, b = optr[i % len(optr)]
                    e2.process_events_batch_device(gptr[i % len(gptr)], offs_a, d, b)
                tm2 = Timer(torch, None, dev, e2.sync)
                est2 = tm2.prewarm(step_aos, PREWARM_S)
                k2 = max(1, args.steps * fps // B)
                R2 = int(min(200, max(3, round(0.2 / max(k2 * est2, 1e-6)))))
                el2, _ = tm2.blocks(lambda: run_steps(step_aos, k2), R2)
                dt = float(np.median(el2))
                step_aos(0)  # group 0 once more: its last frame is one of those kept on the host
                e2.sync()
                other_modes["eventcd_records"] = {"value": round(n_ev * k2 * B / dt / 1e6, 2), "unit": "Mevents/s",
                                                  "ms_per_frame": round(dt / (k2 * B) * 1e3, 5), "blocks": R2, "k1_paths": e2.path_counts(),
                                                  "frames_redone_on_general_path": e2.sorted_fallbacks(), "bytes_per_event_read": 16}
                if B - 1 in host_frames:
                    other_modes["eventcd_records"]["depth_equals_oracle"] = bool(
                        np.array_equal(d2[B - 1].cpu().numpy(), oracle_frame(B - 1, False, False)["depth"]))
                e2.close()
                del A, d2, b2
            except Exception as e:  # never lose the line to an extra leg
                other_modes["eventcd_records"] = {"error": repr(e)[:200]}
        other_modes["note"] = ("eventcd_records = the same groups as 16-byte EventCD records (xm_process_batch_aos), the layout Metavision "
                               "delivers; one_frame_per_call = every frame through its own asynchronous call (xm_process_frame) with "
                               "XM_FLAG_ADAPTIVE_BATCH: a frame that arrives while the GPU is busy is held back and goes out "
                               "with the frames behind it as one set of multi-frame launches (64 slots: groups of up to 16; an idle GPU launches at once); "
                               "one_frame_per_call_eager = the same calls without the flag (three launches per frame, 4 frames in "
                               "flight, a launch thread per slot stream): round 2's headline mode; forced_general = "
                               "XM_FLAG_GENERAL (extrema pass K0 + 64-bit packed keys on every frame, one frame per call: round 1's "
                               "headline mode); camera_view = --camera-perspective; `value` above = library defaults, groups of "
                               f"{B} frames per call" if B else
                               "groups_of_32_frames_per_call = xm_process_batch; launches_from_the_calling_thread = no launch "
                               "workers; forced_general = XM_FLAG_GENERAL; camera_view = --camera-perspective")

    # (these legs run LAST, on an engine of their own: their pinned allocations and extra streams change how the runtime maps
    #  streams to hardware queues for whatever engine comes next -- seen: the following loop at half its rate)
    eng = XMapsEngine(tables, camera_perspective=camera, device=local_rank, n_slots=4, **mode_kw) \
        if (not args.no_host_path and world == 1) else None
    hf4 = [host_frames[f] for f in range(4)]
    leg_clock('other engine settings')
    # ---- PCIe-inclusive figures: events start in host memory, depth + BGR end in host memory (never `value`) ------
    host_path = None
    if not args.no_host_path and world == 1:
        x, y, t = host_frames[0]
        for _ in range(max(3, 4 + 1)):  # every slot allocates its staging buffers on first use
            eng.process_frame(x, y, t, want_bgr=bgr_out is not None)
        c0 = time.perf_counter()
        for _ in range(20):
            eng.process_frame(x, y, t, want_bgr=bgr_out is not None)
        host_path = {"Mevents_per_s_pageable_synchronous": round(20 * n_ev / (time.perf_counter() - c0) / 1e6, 2)}
        pin = []
        for (hx, hy, ht) in hf4:
            px_, py_, pt_ = eng.host_empty(hx.shape, np.uint16), eng.host_empty(hy.shape, np.uint16), eng.host_empty(ht.shape, np.int64)
            px_[:], py_[:], pt_[:] = hx, hy, ht
            pin.append((px_, py_, pt_))
        outs = [(eng.host_empty((H, W), np.float32), None if bgr_out is None else eng.host_empty((H, W, 3), np.uint8))
                for _ in range(4)]
        reps = 200
        for i in range(16):
            a = pin[i % len(pin)]
            eng.process_frame_pinned(a[0], a[1], a[2], None, outs[i % len(outs)][0], outs[i % len(outs)][1])
        eng.sync()
        c0 = time.perf_counter()
        for i in range(reps):
            a = pin[i % len(pin)]
            eng.process_frame_pinned(a[0], a[1], a[2], None, outs[i % len(outs)][0], outs[i % len(outs)][1])
        eng.sync()
        dt = time.perf_counter() - c0
        hf = hf4[(reps - 1) % len(pin)]
        ok_pinned = bool(np.array_equal(outs[(reps - 1) % len(outs)][0],
                                        O.process_ev_frame(tables, hf[0].astype(np.int64), hf[1].astype(np.int64), hf[2],
                                                           camera_perspective=camera, want_bgr=False)["depth"]))
        bytes_per_frame = 12 * n_ev + H * W * (4 + bgr_b)
        host_path.update({"Mevents_per_s_pinned_pipelined": round(reps * n_ev / dt / 1e6, 2),
                          "pcie_GBps": round(reps * bytes_per_frame / dt / 1e9, 2), "depth_equals_oracle": ok_pinned,
                          "meets_north_star_1_Gevent_per_s_end_to_end": bool(reps * n_ev / dt / 1e9 >= 1.0),
                          "note": "end to end: events start in (pinned) host memory, depth+BGR end in host memory, copies of "
                                  "one frame overlap the kernels of another; PCIe-bound; never the headline value"})

    leg_clock('host path')
    # ---- end to end with the device-side ingest: RAW camera packets (all polarities) in host memory -> frames in host memory ----
    ingest_path = None
    if not args.no_host_path and world == 1:
        ingest_path = ingest_leg(eng, hf4, n_ev, O, tables, camera)

    if eng is not None:
        eng.close()
    out = {
        "metric": "Mevents/s to depth frame, 640x480, 1M ev/frame", "value": round(value, 2), "unit": "Mevents/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64+f64",
        "data": "synthetic",
        "config": {"workload": "C-1M: synthetic 1M events/frame, 640x480 cam/proj, rect 1760x1320, 1xMI355X fused kernels"
                   + (" (camera view)" if camera else " (projector view)"),
                   "events_per_frame": n_ev, "frames_per_step": fps, "events_per_step": n_ev * fps, "frames_in_flight": slots,
                   "outputs": "depth f32" + ("" if bgr_out is None else " + BGR u8"),
                   "launch": (f"eager; a step = one group of {B} frames through ONE call (xm_process_batch: one set of multi-frame "
                              f"launches, grid = frames x tiles), {slots // B} groups in flight, launches from the calling thread"
                              if B else ("one frame per call, XM_FLAG_ADAPTIVE_BATCH (frames arriving while the GPU is busy are submitted as groups)"
                                         if adaptive else "eager, one frame per call"
                                         + ("" if args.no_launch_workers else "; XM_FLAG_LAUNCH_WORKERS (a launch thread per slot stream)"))),
                   "inputs": "SoA x:u16 y:u16 t:i64 resident in HBM",
                   "distinct_frames_resident": nf, "resident_set_MB": round(resident_mb, 1),
                   "resident_set_vs_infinity_cache": "exceeds the 256 MiB MALL" if resident_mb > 268.4 else "fits the 256 MiB MALL",
                   "k1_paths_frames": paths,
                   "extrema": "XM_FLAG_GENERAL (K0 every frame)" if args.general else
                              ("XM_FLAG_TIME_SORTED" if args.assume_sorted else
                               "library default: (t[0], t[n-1]) verified on the device; groups take the column-tile K1 (boundary "
                               "pass + plain-store u16 frame), single frames the compact 32-bit key frame; failing frames are "
                               "redone with K0 on the 64-bit path"),
                   "frames_redone_on_general_path": frames_redone},
        "timing": {"prewarm_s": PREWARM_S, "blocks": int(R), "block_s_median": round(elapsed, 6),
                   "block_s_min": round(float(el.min()), 6), "block_s_max": round(float(el.max()), 6),
                   "note": "R blocks of exactly `steps` steps, each bracketed by barrier + synchronize, max over ranks per block, "
                           "median block reported; ms_per_step = one step = frames_per_step frames"},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        "host_enqueue_us_per_step": round(float(np.median(enq)) / args.steps * 1e6, 2),
    }
    if other_modes:
        out["other_modes"] = other_modes
    if host_path:
        out["host_path"] = host_path
    if ingest_path:
        out["ingest_path"] = ingest_path
    if world == 1 and not args.no_other_modes and not args.no_other_configs and B and not camera:
        leg_clock("ingest path")
        out["other_configs"] = other_config_legs(args, torch, dist, dev, local_rank)
        leg_clock("other configs")
    return out


def ingest_leg(eng, host_frames, n_ev, O, tables, camera, n_frames=24):
    """A camera-like stream of C-1M frames (13 ms scans, 3.6 ms dark gaps, 60 Hz) as 1/4-period packets of raw EventCD
    records in PINNED host memory -> xm_ingest_push_pinned -> polarity filter, buffering, pause detection, frame cut, K0/K1/K2
    on the device -> BGR + depth frames in the pinned result ring.  Pushed as fast as the pipeline takes them."""
    from x_maps_amd import synthetic as S
    from x_maps_amd.ingest import DeviceIngest
    from x_maps_amd.trigger_finder import RobustTriggerFinder
    period = 16_600
    total = n_frames * n_ev
    stream = eng.host_empty((total,), S.EVENT_CD_DTYPE)
    for f in range(n_frames):
        x, y, t = host_frames[f % len(host_frames)]
        sl = stream[f * n_ev:(f + 1) * n_ev]
        sl["x"], sl["y"], sl["p"] = x, y, 1
        sl["t"] = t - t[0] + 2_000_000 + f * period
    packet = int(1e6 / 60 / 4)
    edges = np.arange(stream["t"][0], stream["t"][-1] + packet, packet)
    cuts = np.searchsorted(stream["t"], edges)
    cut_frames, first_cut = [], []

    def on_frame(e):
        if not cut_frames:
            first_cut.append(np.array(e))  # the events themselves: (t_first, n) does not say which of several equal stamps is first
        cut_frames.append((int(e["t"][0]), len(e)))
    tf = RobustTriggerFinder(60, on_frame)
    for a, b in zip(cuts[:-1], cuts[1:]):
        tf.process_events(stream[a:b])
    with DeviceIngest(eng, 60, capacity_events=1 << 23, max_packet_events=1 << 20, expected_events_per_frame=n_ev,
                      result_ring=max(8, n_frames)) as ing:
        for a, b in zip(cuts[:-1], cuts[1:]):  # warm-up = the whole stream once, untimed (first launches; one DMA through every ring buffer)
            ing.push_pinned(stream[a:b])
        ing.flush()
        ing.reset()
        ing.poll(copy=False)
        c0 = time.perf_counter()
        for a, b in zip(cuts[:-1], cuts[1:]):
            ing.push_pinned(stream[a:b])
        ing.flush()
        got = ing.poll(copy=False)  # views into the pinned result ring (it holds every frame of this run)
        dt = time.perf_counter() - c0
        hs = ing.host_stats()
        got = [type(f)(**{**f.__dict__, "depth": None if f.depth is None else f.depth.copy(), "bgr": None}) for f in got]  # (kept past the ring)
    same_cut = [(f.t_first, f.n_events) for f in got] == cut_frames
    # the same stream as the recording stores it: EVT 3.0 words (about 4 bytes per event here: every event its own row word),
    # decoded on the device in front of the ingest (xm_ingest_push_evt3) -- a quarter of the bytes cross PCIe
    evt3_leg = None
    try:
        from x_maps_amd import evt3
        evt3_leg = {}
        t_mask = (1 << 24) - 1  # (the format carries 24 bits of time; the stream starts below 2^24 and wraps are counted from 0)
        for label, pk_us in (("quarter_period_chunks", packet), ("period_chunks", 4 * packet)):
            edges3 = np.arange(stream["t"][0], stream["t"][-1] + pk_us, pk_us)
            cuts3 = np.searchsorted(stream["t"], edges3)
            chunks = []
            for a, b in zip(cuts3[:-1], cuts3[1:]):
                if b > a:
                    w = evt3.encode_evt3_singles(stream[a:b])
                    pw = eng.host_empty(w.shape, np.uint16)  # pinned, like the EventCD packets above
                    pw[:] = w
                    chunks.append(pw)
            n_words = int(sum(len(c) for c in chunks))
            with DeviceIngest(eng, 60, capacity_events=1 << 23, max_packet_events=1 << 20, expected_events_per_frame=n_ev,
                              result_ring=max(8, n_frames)) as ing, \
                    evt3.DeviceEvt3Decoder(eng, max_words=max(len(c) for c in chunks)) as dec:
                for c in chunks:  # (warm-up: the whole stream once)
                    dec.push(ing, c, pinned=True, count=False)
                ing.flush()
                ing.reset()
                ing.poll(copy=False)
                dec.reset()
                c0 = time.perf_counter()
                for c in chunks:
                    dec.push(ing, c, pinned=True, count=False)  # nothing waited for: the chunk's event count stays on the device
                ing.flush()
                got3 = ing.poll(copy=False)
                dt3 = time.perf_counter() - c0
                got3 = [type(f)(**{**f.__dict__, "depth": None if f.depth is None else f.depth.copy(), "bgr": None}) for f in got3]
            leg = {"Mevents_per_s_end_to_end": round(total / dt3 / 1e6, 2), "chunks": len(chunks),
                   "bytes_per_event_over_pcie": round(2.0 * n_words / total, 2), "pcie_GBps_in": round(2.0 * n_words / dt3 / 1e9, 2),
                   "frames_cut": len(got3)}
            if label == "quarter_period_chunks":  # the packets of the EventCD run above: the same frames must come out
                leg["same_frames_as_from_eventcd_records"] = bool(
                    [(f.t_first & t_mask, f.n_events) for f in got3] == [(f.t_first & t_mask, f.n_events) for f in got]) and \
                    bool(all(np.array_equal(a.depth, b.depth) for a, b in zip(got3, got)))
            else:  # other packets, other cuts (the trigger finder decides once per packet): every frame against the oracle's cut
                tf3 = RobustTriggerFinder(60, lambda e, acc=leg.setdefault("_cut", []): acc.append((int(e["t"][0]) & t_mask, len(e))))
                for a, b in zip(cuts3[:-1], cuts3[1:]):
                    tf3.process_events(stream[a:b])
                leg["same_frames_as_host_trigger_finder"] = bool([(f.t_first & t_mask, f.n_events) for f in got3] == leg.pop("_cut"))
            evt3_leg[label] = leg
        evt3_leg["note"] = ("the same stream as EVT 3.0 words in pinned host memory -> H2D -> decoded by three scan kernels straight into "
                            "the ingest's packet slot (xm_ingest_push_evt3 with n_events = NULL: nothing is waited for, the ingest's kernels read the "
                            "chunk's event count on the device) -> the same device pipeline; quarter_period_chunks = the packets of the EventCD run, "
                            "period_chunks = one projector period per chunk (an offline replay chooses its chunks)")
    except Exception as e:  # never lose the line to the extra leg
        evt3_leg = {"error": repr(e)[:200]}
    ok = None
    if got and same_cut:
        f0 = got[0]
        ev0 = first_cut[0]  # the host trigger finder's frame (same first stamp and length as the device's: same_cut)
        ref = O.process_ev_frame(tables, ev0["x"].astype(np.int64), ev0["y"].astype(np.int64), np.ascontiguousarray(ev0["t"]),
                                 camera_perspective=camera, want_bgr=False)
        ok = bool(np.array_equal(f0.depth, ref["depth"]))
    return {"Mevents_per_s_end_to_end": round(total / dt / 1e6, 2), "frames_cut": len(got), "frames_in_stream": n_frames,
            "same_frames_as_host_trigger_finder": bool(same_cut), "first_frame_depth_equals_oracle": ok,
            "pcie_GBps_in": round(total * 16 / dt / 1e9, 2), "pushes": hs["pushes"],
            "host_us_per_push": round(hs["us_per_push_without_waits"], 2), "host_us_per_push_incl_backpressure": round(hs["us_per_push"], 2),
            "staging_waits": hs["staging_waits"], "from_evt3_words": evt3_leg,
            "note": "raw 16-byte EventCD packets in pinned host memory -> H2D -> filter / segment / K0-K1-K2 on the device "
                    "(the event stream never returns to the host; three ingest launches + the frame kernels per packet, issued by "
                    "the ingest's launch thread: host_us_per_push is what the calling thread pays) -> depth + BGR in pinned host "
                    "memory, handed out as views into the result ring; pushed back to back, "
                    "i.e. faster than the 60 Hz it was stamped for; the reference's trigger finder cannot cut the first and "
                    "the last frame of a stream"}
