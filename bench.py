#!/usr/bin/env python3
"""bench.py -- Mevents/s to depth frame on synthetic 640x480 / 1 M-event frames (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (K0 minmax -> K1 fused per-event scatter -> K2 frame kernel) over one
C-1M frame (1 000 000 events, camera = projector = 640x480, rectified frame 1760x1320) whose SoA event
columns are already resident in HBM; the result is the f32 depth frame + the BGR u8 frame in HBM.
N > 1: every rank runs the same workload on its own GPU with its own frames (the path shards by frame
with no data-path collective) -> "scaling": "weak"; value = events of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line with the extra objects `roofline` (dominant kernel, HIP-event timed inside
this process) and `cpu_baseline` (the NumPy port of the reference path from oracle/, 1 host core).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# HIP runtime (ROCclr) setting, read when the runtime initialises, i.e. before torch touches the GPU: by default every
# stream is drained by the CPU after each 1000 commands (DEBUG_CLR_MAX_BATCH_SIZE), which showed up as a 1.5-5 ms stall of
# the whole frame pipeline every ~1300 frames and a 7 % slower pace in between (tools/steady_probe.py).
os.environ.setdefault("DEBUG_CLR_MAX_BATCH_SIZE", "100000")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--slots", type=int, default=int(os.environ.get("XM_SLOTS", "4")),
                    help="frames in flight per GPU (key frame + state each; one stream per hardware queue, 4)")
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames resident in HBM")
    ap.add_argument("--camera-perspective", action="store_true")
    ap.add_argument("--no-bgr", action="store_true", help="depth frame only")
    ap.add_argument("--graph", action="store_true", help="replay the K steps from one captured hipGraph")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-path", action="store_true", help="also time the PCIe-inclusive host->host call")
    ap.add_argument("--no-parity", action="store_true", help="EXPERIMENTS ONLY (ablation builds): skip the parity gate")
    ap.add_argument("--assume-sorted", action="store_true",
                    help="XM_FLAG_TIME_SORTED: extrema = t[0], t[n-1], verified on the device; the extrema pass K0 is skipped")
    ap.add_argument("--launch-workers", action="store_true", help="XM_FLAG_LAUNCH_WORKERS: one launch thread per slot stream")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the extra try-sorted / declared-sorted loops")
    ap.add_argument("--try-sorted", action="store_true",
                    help="XM_FLAG_TRY_SORTED: no declaration; (t[0], t[n-1]) tried and verified on every frame, frames that "
                         "fail are redone on the general path automatically")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world

    import torch

    dist = None
    if world > 1 or os.environ.get("XM_BENCH_FORCE_DIST") == "1":  # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from x_maps_amd import XMapsEngine
    from x_maps_amd import synthetic as S

    cfg = S.C_1M
    tables = S.make_tables(cfg)
    eng = XMapsEngine(tables, camera_perspective=args.camera_perspective, device=local_rank, n_slots=args.slots,
                      assume_time_sorted=args.assume_sorted, try_sorted=args.try_sorted,
                      default_priority_streams=args.graph,  # graph replays need default-priority streams (xmaps.h)
                      launch_workers=args.launch_workers)
    H, W = eng.out_h, eng.out_w
    n_ev = cfg.n_events

    # ---- synthetic frames -> HBM (SoA columns, the layout K1 reads) ---------------------------------------
    frames = []
    host_frames = []
    for f in range(args.frames):
        evs = S.make_events(cfg, frame=rank * args.frames + f)
        x, y, t, p = S.to_soa(evs)
        host_frames.append((x, y, t))
        frames.append(tuple(torch.from_numpy(a).to(dev) for a in (x.view(np.int16), y.view(np.int16), t)))
    n_out = max(args.slots, 1) if not args.graph else args.steps
    depth_out = torch.empty((n_out, H, W), dtype=torch.float32, device=dev)
    bgr_out = None if args.no_bgr else torch.empty((n_out, H, W, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step(i):
        fx, fy, ft = frames[i % len(frames)]
        o = i % n_out
        eng.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, n_ev,
                                 depth_out[o].data_ptr(), None if bgr_out is None else bgr_out[o].data_ptr())

    # ---- parity gate before any timing: frame 0 against the CPU oracle (rank 0) ----------------------------
    parity = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import xmaps_oracle as O  # checker + cpu_baseline only
        step(0)
        eng.sync()
        x, y, t = host_frames[0]
        ref = O.process_ev_frame(tables, x.astype(np.int64), y.astype(np.int64), t,
                                 camera_perspective=args.camera_perspective, want_bgr=bgr_out is not None)
        got = depth_out[0].cpu().numpy()
        nz = ref["depth"] != 0
        rel = float((np.abs(got[nz] - ref["depth"][nz]) / ref["depth"][nz]).max(initial=0.0))
        parity = {"depth_max_rel_err": rel, "depth_bit_exact": bool(np.array_equal(got, ref["depth"])),
                  "empty_mask_equal": bool(np.array_equal(got == 0, ref["depth"] == 0))}
        if bgr_out is not None:
            parity["bgr_equal"] = bool(np.array_equal(bgr_out[0].cpu().numpy(), ref["bgr"]))
        st = eng.last_frame_stats()
        parity["n_inliers_equal"] = bool(st.n_inliers == int(ref["mask"].sum()))
        ok = rel <= 1e-4 and parity["empty_mask_equal"] and parity["n_inliers_equal"] and parity.get("bgr_equal", True)
        if not ok and args.no_parity:
            parity["IGNORED"] = True
        elif not ok:
            print(json.dumps({"error": "parity check failed", "parity": parity}))
            sys.exit(1)

    graph = None
    if args.graph:
        # one graph = K frames; events are the resident frames in round-robin order, laid out back to back
        order = [i % len(frames) for i in range(args.steps)]
        gx = torch.cat([frames[i][0] for i in order])
        gy = torch.cat([frames[i][1] for i in order])
        gt = torch.cat([frames[i][2] for i in order])
        offs = np.arange(args.steps + 1, dtype=np.uint64) * n_ev
        torch.cuda.synchronize()
        graph = eng.graph_create(gx.data_ptr(), gy.data_ptr(), gt.data_ptr(), None, offs, depth_out.data_ptr(),
                                 None if bgr_out is None else bgr_out.data_ptr())

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- warmup, then EXACTLY K timed steps bracketed by barrier + synchronize ------------------------------
    if graph is not None:
        for _ in range(max(1, args.warmup // max(args.steps, 1))):
            graph.launch()
    else:
        for i in range(args.warmup):
            step(i)
    barrier()
    t0 = time.perf_counter()
    if graph is not None:
        graph.launch()
    else:
        for i in range(args.steps):
            step(i)
    t_enqueued = time.perf_counter()
    eng.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    total_events = float(n_ev) * args.steps * world
    value = total_events / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events around each kernel on its own stream --------------
        prof = np.zeros((min(args.steps, 200), 4))
        for i in range(len(prof)):
            fx, fy, ft = frames[i % len(frames)]
            st = eng.profile_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, n_ev,
                                          depth_out[0].data_ptr(), None if bgr_out is None else bgr_out[0].data_ptr())
            prof[i] = st.gpu_ms
        k_ms = prof[len(prof) // 10:].mean(axis=0)  # drop the first 10 % (clock ramp)
        rw, rh, pw, ph, cw, ch = (tables[k] for k in ("rect_w", "rect_h", "proj_w", "proj_h", "cam_w", "cam_h"))
        bgr_b = 0 if bgr_out is None else 3
        # algorithmic bytes per launch; K0 is charged nothing (it is an extra pass)
        if args.camera_perspective:
            frame_bytes = (12 + bgr_b) * cw * ch
        else:
            frame_bytes = 8 * rw * rh + (8 + bgr_b) * pw * ph
        alg = {"k_minmax": 0.0, "k_scatter": 24.0 * n_ev, "k_frame": float(frame_bytes)}
        names = ["k_minmax", "k_scatter", "k_frame"]
        dom = int(np.argmax(k_ms[:3]))
        if alg[names[dom]] == 0.0:  # never report the helper pass as the roofline kernel
            dom = 1 if k_ms[1] >= k_ms[2] else 2
        ach = alg[names[dom]] / (k_ms[dom] * 1e-3) / 1e9
        frame_alg = alg["k_scatter"] + alg["k_frame"]
        traffic = None
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_run.sh -> profiles/pmc_traffic.json)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pt = json.load(f)
            wl = "camera" if args.camera_perspective else "projector"
            traffic = pt[wl][names[dom]]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        pipeline_traffic = None
        try:
            tot = sum(pt[wl][k]["hbm_bytes_per_launch"] for k in names)
            pipeline_traffic = {"hbm_bytes_per_frame_all_kernels": tot,
                                "GBps_at_measured_step_time": round(tot / (elapsed / args.steps) / 1e9 / world * world, 1),
                                "frac_of_peak": round(tot / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
        except Exception:
            pass
        roofline = {
            "bound": "hbm", "kernel": names[dom], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
            "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE calibration, MI355X_MICROARCH.md)" if traffic else None,
            "algorithmic_bytes_per_launch": alg[names[dom]],
            "avg_launch_us": {n: round(float(k_ms[i]) * 1e3, 2) for i, n in enumerate(names)},
            "timing": "HIP start/stop events attached to each dispatch (hipExtLaunchKernelGGL) on the stream it runs on",
            "empty_event_pair_us": round(eng.profile_event_overhead_ms(15) * 1e3, 2),
            "frame_us_serial": round(float(k_ms[3]) * 1e3, 2),
            "whole_frame": {"algorithmic_bytes": frame_alg,
                            "achieved_GBps_pipelined": round(frame_alg / (elapsed / args.steps) / 1e9, 2),
                            "frac_of_peak_pipelined": round(frame_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5)},
            "event_stream_read_roofline_frac": round(value * 1e6 / world * 14 / 1e9 / HBM_PEAK_GBS, 5),
            "pipeline_hbm_traffic": pipeline_traffic,
        }

        # ---- CPU baseline: NumPy port of the reference path (same pass structure, 1 core), bounded sample ---
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # CPU baseline: rank 0 at N = 1 only
            x, y, t = host_frames[0]
            xi, yi = x.astype(np.int64), y.astype(np.int64)
            reps, spent = 0, 0.0
            best = 1e9
            while spent < args.cpu_seconds and reps < 50:
                c0 = time.perf_counter()
                O.process_ev_frame(tables, xi, yi, t, camera_perspective=args.camera_perspective,
                                   want_bgr=bgr_out is not None)
                dt = time.perf_counter() - c0
                best = min(best, dt)
                spent += dt
                reps += 1
            cpu = {"value": round(n_ev / (spent / reps) / 1e6, 3), "unit": "Mevents/s", "cores": 1, "kind": "port",
                   "sample": f"{reps} x the C-1M frame 0 (1 M events -> depth+BGR), mean; best {n_ev / best / 1e6:.2f} Mev/s; "
                             "NumPy port with the reference's pass structure (its per-event path is 1-threaded NumPy)",
                   "host_cpus": os.cpu_count()}
            try:  # upper bound for the reference: fused C loops on every host core (what Numba prange could reach)
                from c_oracle import COracle
                co = COracle(tables, args.camera_perspective, omp=True)
                co.process_ev_frame(x, y, t, want_events=False)
                c0 = time.perf_counter()
                creps = 0
                while time.perf_counter() - c0 < min(3.0, args.cpu_seconds) and creps < 200:
                    co.process_ev_frame(x, y, t, want_events=False)
                    creps += 1
                cdt = (time.perf_counter() - c0) / creps
                cpu["all_cores_c_openmp"] = {"value": round(n_ev / cdt / 1e6, 2), "unit": "Mevents/s",
                                             "cores": co.threads, "kind": "port", "sample": f"{creps} x frame 0"}
            except Exception as e:  # the checker is optional for the bench
                cpu["all_cores_c_openmp"] = {"error": str(e)[:200]}

        host_path = None
        if args.host_path:
            x, y, t = host_frames[0]
            for _ in range(max(3, args.slots + 1)):  # every slot allocates its staging buffers on first use
                eng.process_frame(x, y, t, want_bgr=bgr_out is not None)
            c0 = time.perf_counter()
            for _ in range(20):
                eng.process_frame(x, y, t, want_bgr=bgr_out is not None)
            host_path = {"Mevents_per_s_pageable_synchronous": round(20 * n_ev / (time.perf_counter() - c0) / 1e6, 2)}
            # pinned host buffers, asynchronous, copies of one frame overlapping the kernels of another (n_slots streams)
            pin = []
            for (hx, hy, ht) in host_frames[:4]:
                px_, py_, pt_ = eng.host_empty(hx.shape, np.uint16), eng.host_empty(hy.shape, np.uint16), eng.host_empty(ht.shape, np.int64)
                px_[:], py_[:], pt_[:] = hx, hy, ht
                pin.append((px_, py_, pt_))
            outs = [(eng.host_empty((H, W), np.float32), None if bgr_out is None else eng.host_empty((H, W, 3), np.uint8))
                    for _ in range(max(args.slots, 1))]
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
            ok_pinned = bool(np.array_equal(outs[(reps - 1) % len(outs)][0],
                                            O.process_ev_frame(tables, *[v.astype(np.int64) if v.dtype != np.int64 else v
                                                                         for v in host_frames[(reps - 1) % len(pin)]],
                                                               camera_perspective=args.camera_perspective, want_bgr=False)["depth"]))
            bytes_per_frame = 12 * n_ev + H * W * (4 + (0 if bgr_out is None else 3))
            host_path.update({"Mevents_per_s_pinned_pipelined": round(reps * n_ev / dt / 1e6, 2),
                              "pcie_GBps": round(reps * bytes_per_frame / dt / 1e9, 2), "depth_equals_oracle": ok_pinned,
                              "note": "events start in (pinned) host memory, depth+BGR end in host memory; never the headline value"})

        # ---- the same loop in the engine's other extrema modes (extra information, never the headline `value`) ----
        other_modes = None
        if graph is not None:
            graph.close()
            graph = None
        frames_redone = eng.sorted_fallbacks()
        eng.close()  # one engine at a time: two engines share the high-priority hardware queues
        eng = None
        if world == 1 and not args.graph and not args.assume_sorted and not args.try_sorted and not args.no_other_modes:
            other_modes = {}
            modes = [("try_sorted", {"try_sorted": True}), ("declared_sorted", {"assume_time_sorted": True})]
            if os.environ.get("XM_BENCH_GENERAL_AGAIN"):  # experiment: the headline mode measured again on a second engine
                modes = [("general_again", {})] + modes + [("general_again2", {})]
            for name, kw in modes:
                e2 = XMapsEngine(tables, camera_perspective=args.camera_perspective, device=local_rank, n_slots=args.slots, **kw)

                def step2(i):
                    fx, fy, ft = frames[i % len(frames)]
                    o = i % n_out
                    e2.process_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, n_ev, depth_out[o].data_ptr(),
                                            None if bgr_out is None else bgr_out[o].data_ptr())
                for i in range(args.warmup):
                    step2(i)
                e2.sync()
                c0 = time.perf_counter()
                for i in range(args.steps):
                    step2(i)
                e2.sync()
                dt = time.perf_counter() - c0
                same = bool(np.array_equal(depth_out[(args.steps - 1) % n_out].cpu().numpy(),
                                           O.process_ev_frame(tables, *[v.astype(np.int64) if v.dtype != np.int64 else v
                                                                        for v in host_frames[(args.steps - 1) % len(frames)]],
                                                              camera_perspective=args.camera_perspective, want_bgr=False)["depth"]))
                other_modes[name] = {"value": round(n_ev * args.steps / dt / 1e6, 2), "unit": "Mevents/s",
                                     "ms_per_step": round(dt / args.steps * 1e3, 5), "depth_equals_oracle": same,
                                     "frames_redone_on_general_path": e2.sorted_fallbacks()}
                e2.close()
            other_modes["note"] = ("try_sorted = XM_FLAG_TRY_SORTED (no declaration: (t[0], t[n-1]) tried and verified on the device, "
                                   "failing frames redone automatically); declared_sorted = XM_FLAG_TIME_SORTED; `value` above is "
                                   "the general path (extrema pass K0 on every frame)")

        out = {
            "metric": "Mevents/s to depth frame, 640x480, 1M ev/frame", "value": round(value, 2), "unit": "Mevents/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64+f64",
            "data": "synthetic",
            "config": {"workload": "C-1M: synthetic 1M events/frame, 640x480 cam/proj, rect 1760x1320, 1xMI355X fused kernels"
                       + (" (camera view)" if args.camera_perspective else " (projector view)"),
                       "events_per_frame": n_ev, "frames_in_flight": args.slots, "outputs": "depth f32" + ("" if bgr_out is None else " + BGR u8"),
                       "launch": "hipGraph" if args.graph else "eager", "inputs": "SoA x:u16 y:u16 t:i64 resident in HBM",
                       "time_sorted_declared": bool(args.assume_sorted), "try_sorted": bool(args.try_sorted),
                       "frames_redone_on_general_path": frames_redone},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
            "host_enqueue_us_per_step": round((t_enqueued - t0) / args.steps * 1e6, 2),
        }
        if other_modes:
            out["other_modes"] = other_modes
        if host_path:
            out["host_path"] = host_path
        print(json.dumps(out))
    if graph is not None:
        graph.close()
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
