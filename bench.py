#!/usr/bin/env python3
"""bench.py -- Mevents/s to depth frame on synthetic 640x480 / 1 M-event frames (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1] (C-1M): a "step" = one pass of the hot path over one batch of synthetic input =
ONE GROUP of 32 frames of 1 000 000 events each (camera = projector = 640x480, rectified frame 1760x1320) through one
xm_process_batch call, i.e. one set of multi-frame launches (boundary pass K0b -> K1 column tiles -> K2 frame kernel, grid =
frames x tiles; K0 + the 64-bit key path only for frames whose verified (t[0], t[n-1]) shortcut fails).  The frames' SoA
event columns are already resident in HBM; the result is the f32 depth frame + the BGR u8 frame per frame in HBM.  The engine
runs with the library's default flags.  `--batch 0`: a step = one frame through one asynchronous call (round 2's mode).
N > 1: every rank runs the same workload on its own GPU with its own frames (the path shards by frame with no data-path
collective) -> "scaling": "weak"; value = events of all ranks / max-over-ranks time.

How the K steps are timed: W warm-up steps, then a fixed wall-clock pre-warm (PREWARM_S, so that clocks and caches are in
the same state whatever K and W are), then R blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both
sides; per block the MAX over ranks is taken and the MEDIAN block is reported (R is chosen so that the blocks together
last about TARGET_TIMED_S; R, and the fastest / slowest block, are in the JSON line).  ms_per_step = median block / K.

Other workloads (never mixed into the default line):
    --graph    BASELINE configs[4]: 60 frames x 1 M events replayed from one captured hipGraph: throughput + latency
    --sharded  BASELINE configs[3]: C-10M (1280x720, 10 M events/frame) sharded by event index over the ranks, packed-key
               frame MAX-all-reduced over RCCL; value = events / max-rank time, collective time reported separately

Rank 0 prints ONE JSON line with the extra objects `roofline` (dominant kernel, timed with HIP events attached to its own
dispatch, BEFORE the timed blocks) and `cpu_baseline` (the NumPy port of the reference path from oracle/, 1 host core).
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
PREWARM_S = float(os.environ.get("XM_BENCH_PREWARM_S", "0.35"))  # wall-clock pre-warm before anything is measured
TARGET_TIMED_S = 0.30  # the R timed blocks together


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps per block; a step = one group of --batch frames")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--slots", type=int, default=int(os.environ.get("XM_SLOTS", "0")),
                    help="frames in flight per GPU (key frame + state each); default 4 (one stream per hardware queue), "
                         "60 with --graph")
    ap.add_argument("--frames", type=int, default=0,
                    help="distinct synthetic frames resident in HBM; default: one distinct group per group in flight (4 x 32 frames = "
                         "1.5 GB of events: no group re-reads what another one has just pulled into the 256 MiB Infinity Cache), "
                         "32 with --batch 0")
    ap.add_argument("--camera-perspective", action="store_true")
    ap.add_argument("--no-bgr", action="store_true", help="depth frame only")
    ap.add_argument("--graph", action="store_true", help="config 5: 60 x C-1M frames replayed from one captured hipGraph")
    ap.add_argument("--sharded", action="store_true", help="config 4: C-10M sharded by event index over the ranks (RCCL)")
    ap.add_argument("--esl", action="store_true",
                    help="configs 1/3 stand-in: ESL-like frames (real calibration geometry, ~150 k events, projector 1080x1920)")
    ap.add_argument("--merge", choices=("columns", "all_reduce", "reduce_scatter", "bands"), default="columns",
                    help="--sharded: how the shards are merged (x_maps_amd/sharded.py).  columns (default): every time column on one rank, "
                         "plain u16 frames merged by SUM (falls back to all_reduce on rigs that do not take the column tiles); the others: "
                         "packed 64-bit keys merged by MAX")
    ap.add_argument("--esl-stream-child", action="store_true",
                    help=argparse.SUPPRESS)  # (internal: --esl's stream legs once more in a process that never imports torch)
    ap.add_argument("--comm", choices=("library", "torch"), default="library",
                    help="--sharded: who issues the collectives.  library (default): xm_shard_comm_* -- the library owns an RCCL "
                         "communicator per lane and one native call per frame enqueues kernels and collectives (merges: columns, "
                         "all_reduce); torch: x_maps_amd.sharded.ShardedFrameProcessor over torch.distributed (every merge; also "
                         "the fall-back when the library's communicator cannot be set up on every rank)")
    ap.add_argument("--lanes", type=int, default=2, choices=(1, 2, 4),
                    help="--sharded: frames in flight.  Each lane is an engine of its own (stream, frame buffers, helper slots) "
                         "taking every lanes-th frame: one frame's latency-bound ends (pack, all-gather, boundary pass; all-reduce, "
                         "K2) run beside the next frame's K1.  1 = one frame at a time (rounds 1-3)")
    ap.add_argument("--batch", type=int, default=32,
                    help="frames per step: a step = ONE group of B C-1M frames through xm_process_batch (one set of multi-frame "
                         "launches, grid = frames x tiles); 0 = a step is one frame through one asynchronous call (round 2's "
                         "headline mode, reported under other_modes by default)")
    ap.add_argument("--groups-in-flight", type=int, default=4,
                    help="with --batch: slots = groups x B (default 4 = one group per stream / hardware queue of the handle: 149-151 -> "
                         "162-166 Gev/s against 3 in alternating runs, 5 / 6 / 8 no different)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive host->host figures")
    ap.add_argument("--no-parity", action="store_true", help="EXPERIMENTS ONLY (ablation builds): skip the parity gate")
    ap.add_argument("--general", action="store_true", help="XM_FLAG_GENERAL: extrema pass K0 on every frame")
    ap.add_argument("--assume-sorted", action="store_true",
                    help="XM_FLAG_TIME_SORTED: extrema = t[0], t[n-1], verified on the device, violations reported")
    ap.add_argument("--no-adaptive", action="store_true",
                    help="one frame per call WITHOUT XM_FLAG_ADAPTIVE_BATCH (every frame its own three launches, 4 in flight)")
    ap.add_argument("--no-launch-workers", action="store_true",
                    help="launch from the calling thread (default: XM_FLAG_LAUNCH_WORKERS, one launch thread per slot stream -- the "
                         "two kernel launches of a frame cost a Python caller ~10 us, about what the GPU needs for the frame)")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the extra loops (forced general, declared sorted, ...)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact legs of the other BASELINE configs (esl, graph60, sharded_c10m)")
    ap.add_argument("--lib-option", action="append", default=[], metavar="NAME=VALUE",
                    help="variant switch of the library for experiments (xm_debug_option), e.g. XM_COLS=0, XM_K2_PIPE=0; repeatable")
    ap.add_argument("--single-block", action="store_true", help="one timed block of K steps (no repetition)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two short child runs under rocprofv3 --pmc, ~30 s); the committed "
                         "profiles/pmc_traffic.json is quoted instead, with its age")
    return ap.parse_args()


class Timer:
    """R blocks of exactly K steps, each bracketed by barrier + synchronize; MAX over ranks per block; median block."""

    def __init__(self, torch, dist, dev, sync):
        self.torch, self.dist, self.dev, self.sync = torch, dist, dev, sync

    def barrier(self):
        self.sync()
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def prewarm(self, step_fn, seconds):
        """Run step_fn(i) pipelined for `seconds` of wall time; returns the observed seconds per step."""
        self.barrier()
        t0 = time.perf_counter()
        i = 0
        while True:
            for _ in range(64):
                step_fn(i)
                i += 1
            if time.perf_counter() - t0 >= seconds:
                break
        self.sync()
        self.torch.cuda.synchronize()
        return (time.perf_counter() - t0) / i

    def agree(self, value):
        """MAX over ranks of a host scalar (every rank must derive the same number of timed blocks from it)."""
        if self.dist is None:
            return value
        tt = self.torch.tensor([value], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def blocks(self, run_block, n_blocks):
        """run_block() enqueues exactly K steps; returns (per-block seconds [max over ranks], per-block host enqueue s)."""
        el, enq = [], []
        for _ in range(n_blocks):
            self.barrier()
            t0 = time.perf_counter()
            run_block()
            t_enq = time.perf_counter()
            self.sync()
            self.torch.cuda.synchronize()
            t1 = time.perf_counter()
            el.append(t1 - t0)
            enq.append(t_enq - t0)
        if self.dist is not None:
            self.dist.barrier()
            tt = self.torch.tensor(el, dtype=self.torch.float64, device=self.dev)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            el = [float(v) for v in tt.cpu()]
        return np.array(el), np.array(enq)


def n_blocks_for(args, est_step_s):
    if args.single_block:
        return 1
    return int(min(400, max(3, round(TARGET_TIMED_S / max(args.steps * est_step_s, 1e-6)))))


def depth_parity(got, ref_depth):
    nz = ref_depth != 0
    rel = float((np.abs(got[nz] - ref_depth[nz]) / ref_depth[nz]).max(initial=0.0))
    return {"depth_max_rel_err": rel, "depth_bit_exact": bool(np.array_equal(got, ref_depth)),
            "empty_mask_equal": bool(np.array_equal(got == 0, ref_depth == 0))}


def cpu_baseline_leg(args, O, tables, host_frame, n_ev, camera, want_bgr):
    """NumPy port of the reference path (same pass structure, 1 core) on a bounded sample + the C/OpenMP port on all cores."""
    x, y, t = host_frame
    xi, yi = x.astype(np.int64), y.astype(np.int64)
    reps, spent, best = 0, 0.0, 1e9
    while spent < args.cpu_seconds and reps < 400:
        c0 = time.perf_counter()
        O.process_ev_frame(tables, xi, yi, t, camera_perspective=camera, want_bgr=want_bgr)
        dt = time.perf_counter() - c0
        best = min(best, dt)
        spent += dt
        reps += 1
    cpu = {"value": round(n_ev / (spent / reps) / 1e6, 3), "unit": "Mevents/s", "cores": 1, "kind": "port",
           "sample": f"{reps} x frame 0 of the workload ({n_ev} events -> depth{'+BGR' if want_bgr else ''}), mean; best "
                     f"{n_ev / best / 1e6:.2f} Mev/s; NumPy port with the reference's pass structure (its per-event path is "
                     "1-threaded NumPy)",
           "host_cpus": os.cpu_count()}
    try:  # upper bound for the reference: fused C loops on every host core (what Numba prange could reach)
        from c_oracle import COracle
        co = COracle(tables, camera, omp=True)
        co.process_ev_frame(x, y, t, want_events=False)
        c0 = time.perf_counter()
        creps = 0
        while time.perf_counter() - c0 < min(3.0, args.cpu_seconds) and creps < 200:
            co.process_ev_frame(x, y, t, want_events=False)
            creps += 1
        cdt = (time.perf_counter() - c0) / creps
        cpu["all_cores_c_openmp"] = {"value": round(n_ev / cdt / 1e6, 2), "unit": "Mevents/s", "cores": co.threads,
                                     "kind": "port", "sample": f"{creps} x frame 0"}
    except Exception as e:  # the checker is optional for the bench
        cpu["all_cores_c_openmp"] = {"error": str(e)[:200]}
    return cpu


def roofline_dict(k_ms, prof, n_ev, B, tables, camera, bgr_b, wl, timing, cell_bytes=2):
    """The `roofline` object from per-launch kernel durations (ms): k_ms = (helper pass, K1, K2, whole step).
    n_ev = events per frame, B = frames per launch.  Three yardsticks side by side for the dominant kernel (never the helper
    pass): SURVEY 8(d)'s algorithmic bytes (`frac`), the HBM bytes the PMC counters saw (`frac_counter_bytes`, from the committed
    rocprofv3 passes of this workload: profiles/pmc_traffic.json[wl]), and -- filled in by pipeline_fractions() once the
    pipelined rate is known -- the event stream's 14 B/event against the HBM read peak."""
    rw, rh, pw, ph, cw, ch = (tables[k] for k in ("rect_w", "rect_h", "proj_w", "proj_h", "cam_w", "cam_h"))
    # algorithmic bytes per launch; the helper pass is charged nothing (it is an extra pass)
    frame_bytes = (12 + bgr_b) * cw * ch if camera else 8 * rw * rh + (8 + bgr_b) * pw * ph
    alg = {"k_minmax": 0.0, "k_scatter": 24.0 * n_ev * B, "k_frame": float(frame_bytes) * B}
    # what the frame kernel cannot avoid moving with the cell format it reads today (u16 / u32 / u64 cells, never cleared): one
    # read of the disparity frame + its per-pixel patch offsets (u32) + the outputs
    k2_min = ((cell_bytes + 4 + bgr_b) * cw * ch if camera else cell_bytes * rw * rh + (4 + 4 + bgr_b) * pw * ph) * B
    names = ["k_minmax", "k_scatter", "k_frame"]
    dom = 1 if k_ms[1] >= k_ms[2] else 2
    pt = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
    except Exception:
        pt = None

    def traffic_of(name):
        try:
            e = pt[wl][name]
            return int(e["hbm_bytes_per_frame"] * B) if "hbm_bytes_per_frame" in e else int(e["hbm_bytes_per_launch"])
        except Exception:
            return None

    kernels = {}
    for i, nme in enumerate(names):
        if k_ms[i] <= 0:
            continue
        t_s = k_ms[i] * 1e-3
        tr = traffic_of(nme)
        kernels[nme] = {"avg_launch_us": round(float(k_ms[i]) * 1e3, 2), "us_per_frame": round(float(k_ms[i]) * 1e3 / B, 3),
                        "algorithmic_bytes_per_launch": alg[nme],
                        "frac_algorithmic": round(alg[nme] / t_s / 1e9 / HBM_PEAK_GBS, 5),
                        "hbm_bytes_per_launch_counters": tr,
                        "frac_counter_bytes": None if tr is None else round(tr / t_s / 1e9 / HBM_PEAK_GBS, 5)}
        if nme == "k_scatter":
            kernels[nme]["frac_event_stream_read"] = round(14.0 * n_ev * B / t_s / 1e9 / HBM_PEAK_GBS, 5)
        if nme == "k_frame":
            kernels[nme]["own_minimal_bytes_per_launch"] = float(k2_min)
            kernels[nme]["frac_own_minimal_bytes"] = round(k2_min / t_s / 1e9 / HBM_PEAK_GBS, 5)
    ach = alg[names[dom]] / max(k_ms[dom] * 1e-3, 1e-12) / 1e9
    traffic = traffic_of(names[dom])
    out = {
        "bound": "hbm", "kernel": names[dom], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
        "frac_counter_bytes": None if traffic is None else round(traffic / (k_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "traffic_source": (f"profiles/pmc_traffic.json[{wl}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, "
                           "2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE calibration, MI355X_MICROARCH.md)") if traffic else None,
        "algorithmic_bytes_per_launch": alg[names[dom]], "frames_per_launch": B,
        "fractions_note": "frac = SURVEY 8(d) algorithmic bytes of the dominant kernel / its launch time / 8 TB/s; frac_counter_bytes = "
                          "the HBM bytes the counters saw instead; event_stream_read_roofline_frac = 14 B/event at the pipelined rate "
                          "against the HBM read peak (the north star's yardstick: 571 Gev/s = 1.0)",
        "kernels": kernels,
        "avg_launch_us": {n: round(float(k_ms[i]) * 1e3, 2) for i, n in enumerate(names)},
        "timing": timing,
    }
    if prof is not None:
        out["launch_us_p10_p90"] = {n: [round(float(np.percentile(prof[:, i], q)) * 1e3, 2) for q in (10, 90)] for i, n in enumerate(names)}
    return out, alg, pt


def pipeline_fractions(roofline, alg, pt, wl, value, world, s_frame, frames_per_launch, helper_runs=True):
    """Whole-pipeline figures at the measured (pipelined) seconds per frame."""
    frame_alg = (alg["k_scatter"] + alg["k_frame"]) / frames_per_launch
    roofline["whole_frame"] = {"algorithmic_bytes": frame_alg,
                               "achieved_GBps_pipelined": round(frame_alg / s_frame / 1e9, 2),
                               "frac_of_peak_pipelined": round(frame_alg / s_frame / 1e9 / HBM_PEAK_GBS, 5)}
    roofline["event_stream_read_roofline_frac"] = round(value * 1e6 / world * 14 / 1e9 / HBM_PEAK_GBS, 5)
    try:
        names = ["k_scatter", "k_frame"] + (["k_minmax"] if helper_runs else [])
        tot = sum(pt[wl][k].get("hbm_bytes_per_frame", pt[wl][k].get("hbm_bytes_per_launch")) for k in names)
        roofline["pipeline_hbm_traffic"] = {"hbm_bytes_per_frame_all_kernels": tot, "kernels": names,
                                            "GBps_at_measured_step_time": round(tot / s_frame / 1e9, 1),
                                            "frac_of_peak": round(tot / s_frame / 1e9 / HBM_PEAK_GBS, 4)}
    except Exception:
        pass


def traffic_file_age():
    """how old the committed counter bytes are (profiles/pmc_traffic.json: the last commit that touched it, else the file's mtime)"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        import subprocess
        r = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%cs %h", "--", "profiles/pmc_traffic.json"], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return "committed " + r.stdout.strip()
    except Exception:
        pass
    try:
        return "file dated " + time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(path)))
    except Exception:
        return "unknown"


def apply_measured_traffic(roofline, measured, detail, s_frame, frames_per_launch):
    """roofline.traffic & co. from counters collected in THIS run (benchmodes/pmc.py) instead of the committed file"""
    if not measured:
        roofline["traffic_measured_in_run"] = False
        roofline["traffic_in_run_note"] = (detail or {}).get("error", "not attempted")
        if roofline.get("traffic_source"):
            roofline["traffic_source"] += "; NOT re-measured in this run (" + roofline["traffic_in_run_note"] + "): " + traffic_file_age()
        return
    for name, k in roofline["kernels"].items():
        if name in measured:
            t_s = k["avg_launch_us"] * 1e-6
            k["hbm_bytes_per_launch_counters"] = measured[name]
            k["frac_counter_bytes"] = round(measured[name] / t_s / 1e9 / HBM_PEAK_GBS, 5)
            k["counters"] = detail["kernels"].get(name)
    dom = roofline["kernel"]
    if dom in measured:
        roofline["traffic"] = measured[dom]
        roofline["frac_counter_bytes"] = roofline["kernels"][dom]["frac_counter_bytes"]
    roofline["traffic_source"] = detail["source"]
    roofline["traffic_measured_in_run"] = True
    roofline["traffic_measure_seconds"] = detail["seconds"]
    tot = sum(measured.get(k, 0) for k in ("k_minmax", "k_scatter", "k_frame")) / frames_per_launch
    roofline["pipeline_hbm_traffic"] = {"hbm_bytes_per_frame_all_kernels": int(tot), "kernels": sorted(measured),
                                        "GBps_at_measured_step_time": round(tot / s_frame / 1e9, 1),
                                        "frac_of_peak": round(tot / s_frame / 1e9 / HBM_PEAK_GBS, 4)}


def roofline_of(eng, frames, n_ev, outs, tables, camera, bgr_b, world, group=None, wl_suffix="", cell_bytes=None):
    """Per-kernel launch durations from HIP events attached to each dispatch.  One frame per launch: 300 serial frames, median
    of the last 200.  group = (B, call): 60 serial groups of B frames (multi-frame launches), median of the last 40."""
    B = group[0] if group else 1
    n_prof, skip = (60, 20) if group else (300, 100)
    prof = np.zeros((n_prof, 4))
    for i in range(n_prof):
        if group:
            prof[i] = group[1](i)
        else:
            fx, fy, ft = frames[i % len(frames)]
            st = eng.profile_frame_device(fx.data_ptr(), fy.data_ptr(), ft.data_ptr(), None, n_ev, outs[0], outs[1])
            prof[i] = st.gpu_ms
    k_ms = np.median(prof[skip:], axis=0)
    wl = ("camera" if camera else "projector") + ("_groups" if group else "") + wl_suffix
    timing = ("HIP start/stop events attached to each dispatch (hipExtLaunchKernelGGL) on the stream it runs on; "
              + (f"{n_prof} serial groups of {B} frames (multi-frame launches, grid = frames x tiles) after the pre-warm and "
                 f"BEFORE the timed blocks, median of the last {n_prof - skip}; k_minmax = the helper pass in front of K1: the "
                 "boundary pass k_cols_bounds of the column-tile / owner-tile path (or the extrema pass K0 on the general path)"
                 if group else
                 "300 serial frames after the pre-warm and BEFORE the timed blocks, median of the last 200; k_minmax = 0: "
                 "not launched (verified (t[0], t[n-1]) shortcut)"))
    if cell_bytes is None:
        cell_bytes = 2 if group else 4
    r, alg, pt = roofline_dict(k_ms, prof[skip:], n_ev, B, tables, camera, bgr_b, wl, timing, cell_bytes)
    r["empty_event_pair_us"] = round(eng.profile_event_overhead_ms(15) * 1e3, 2)
    r["group_us_serial" if group else "frame_us_serial"] = round(float(k_ms[3]) * 1e3, 2)
    return r, alg, pt, wl


def parity_ok(p):
    """every boolean of a parity record true, every relative error within the north star's 1e-4"""
    if p is None:
        return None
    if isinstance(p, bool):
        return p
    if isinstance(p, dict):
        ok = True
        for k, v in p.items():
            if k.endswith("rel_err") and isinstance(v, (int, float)):
                ok = ok and v <= 1e-4
            elif isinstance(v, (bool, dict)):
                r = parity_ok(v)
                ok = ok and (r is not False)
        return ok
    return None


def other_config_legs(args, torch, dist, dev, local_rank):
    """The other BASELINE configs, compact, inside the default line (the driver runs only that one): configs[0]/[2] stand-in
    (--esl, groups of 32 ESL-like frames + the camera-like stream through the device ingest), configs[4] (--graph) and configs[3]
    (--sharded, on this one rank).  Each leg is the corresponding bench mode with fewer steps and without its own extra legs."""
    import copy
    legs = {}

    def compact(out, seconds):
        r = out.get("roofline") or {}
        leg = {"value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"], "steps": out["steps"],
               "workload": out["config"]["workload"][:110],
               "roofline": {k: r[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "frac_counter_bytes", "traffic", "k_scatter_brackets",
                                                  "frac_k1_alone") if k in r},
               "parity_ok": parity_ok(out.get("parity")), "leg_seconds": round(seconds, 1)}
        if "frames_per_step" in out["config"]:
            leg["frames_per_step"] = out["config"]["frames_per_step"]
        if "us_per_frame" in out["config"]:
            leg["us_per_frame"] = out["config"]["us_per_frame"]
        if "latency_us" in out:
            leg["latency_us"] = {k: v for k, v in out["latency_us"].items() if k != "definition"}
        if "collective_ms" in out:
            leg["collective_ms"] = out["collective_ms"]
            for k in ("merge", "fell_back", "frames_in_flight", "collectives_issued_by", "comm_note", "Mevents_per_s_via_torch_distributed",
                      "Mevents_per_s_one_frame_at_a_time"):
                if k in out["config"]:
                    leg[k] = out["config"][k]
        ip = out.get("ingest_path")
        ing_keys = ("Mevents_per_s_end_to_end", "frames_per_s", "ms_per_cut_frame", "frames_cut", "activity_filter",
                    "same_frames_as_host_trigger_finder", "first_frame_equals_oracle", "host_us_per_push", "outputs",
                    "chunks_judged_sequentially", "overflow", "error")
        if isinstance(ip, dict):
            leg["ingest_path"] = {k: ip[k] for k in ing_keys if k in ip}
        if "per_frame_host_call_ms" in out:  # (configs[0]: one ESL-like frame through process_ev_frame's call, host to host)
            leg["per_frame_host_call_ms"] = {k: v for k, v in out["per_frame_host_call_ms"].items() if k != "definition"}
        if out.get("cpu_baseline"):
            leg["cpu_baseline"] = out["cpu_baseline"]
        sl = out.get("stream_legs")
        if isinstance(sl, dict):
            for k in ("ingest_path_filter_off", "from_evt3_words_period_chunks", "from_evt3_words_period_chunks_filter_off"):
                if isinstance(sl.get(k), dict):
                    leg[k] = {a: sl[k][a] for a in ing_keys if a in sl[k]}
            if isinstance(sl.get("paced"), dict):
                leg["paced"] = sl["paced"]
            for k in ("full_replay_through_processor_host_trigger_finder", "full_replay_through_processor_device_ingest"):
                if isinstance(sl.get(k), dict):
                    leg[k] = {a: b for a, b in sl[k].items() if a != "note"}
            ch = sl.get("in_a_process_without_torch")
            if isinstance(ch, dict):  # (the ingest leg and the processor's device-ingest leg: the two the reference's application runs)
                leg["in_a_process_without_torch"] = {k: ch[k] for k in ("ingest_path", "full_replay_through_processor_device_ingest", "error") if k in ch}
        return leg

    plan = (("esl", bench_esl, dict(steps=10, esl=True, no_host_path=False)),
            ("graph60", bench_graph, dict(steps=120, graph=True, no_host_path=True, slots=0, frames=0)),
            ("sharded_c10m", bench_sharded, dict(steps=20, sharded=True, no_host_path=True, slots=0, frames=0)))
    for name, fn, over in plan:
        a = copy.copy(args)
        a.no_cpu_baseline, a.no_other_modes, a.single_block, a.batch, a.groups_in_flight = True, True, False, 32, 4
        if name == "esl":  # (configs[0] is "single frame, CPU reference path": the port timed on an ESL-like frame, ~5 s)
            a.no_cpu_baseline, a.cpu_seconds = False, min(args.cpu_seconds, 4.0)
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        try:
            d = dist
            if name == "sharded_c10m" and d is None:
                import tempfile
                import torch.distributed as d
                d.init_process_group("nccl", init_method=f"file://{tempfile.mkdtemp()}/rdzv", rank=0, world_size=1,
                                     device_id=torch.device("cuda", local_rank))
            try:
                legs[name] = compact(fn(a, torch, d, dev, 0, local_rank, 1), time.perf_counter() - t0)
            finally:
                if name == "sharded_c10m" and dist is None:
                    d.destroy_process_group()
        except BaseException as e:  # (a leg's parity gate exits: never lose the default line to it)
            legs[name] = {"error": repr(e)[:300]}
    return legs


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: this process becomes the launcher -- it re-executes the
    script under torch.distributed.run with one rank per GPU (the command line the driver uses) and hands back its exit code;
    rank 0 of that run prints the one JSON line.  Fewer than N GPUs visible: an error line and a non-zero exit, never a silent
    N = 1 run."""
    import socket
    import subprocess
    n = args.gpus
    if os.environ.get("XM_BENCH_DRY") != "1":
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print(json.dumps({"error": f"--gpus {n} but only {have} GPU(s) visible", "n_gpus_requested": n, "n_gpus_visible": have}), flush=True)
            sys.exit(2)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, XM_BENCH_SPAWNED="1")
    sys.stderr.write("[bench] --gpus %d without a launcher: %s\n" % (n, " ".join(cmd)))
    sys.exit(subprocess.call(cmd, env=env))


def dry_run(args, rank, world):
    """XM_BENCH_DRY=1 (CPU tests of the launch logic): the ranks meet over gloo, count each other with an all-reduce, rank 0
    prints a line with the contract's keys and no measurement."""
    import torch
    import torch.distributed as dist
    seen = 1
    out = {"metric": "dry run (no GPU work)", "value": 0.0, "unit": "Mevents/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 0.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int64+f64", "data": "synthetic", "config": {"workload": "dry"}, "ranks_seen": seen,
           "spawned_by_bench": os.environ.get("XM_BENCH_SPAWNED") == "1"}
    leg_failed = False
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        out["ranks_seen"] = int(t.item())
        dist.barrier()
        # XM_BENCH_DRY_LEG = "ok" | "hang:<rank>" | "die:<rank>": a stand-in for the sharded leg behind the replicas -- two
        # collectives, between which the named rank goes to sleep for ever / leaves -- under the very guard the real leg runs under
        mode = os.environ.get("XM_BENCH_DRY_LEG")
        if mode:
            kind, _, who = mode.partition(":")

            def fake_leg():
                dist.all_reduce(torch.ones(1))
                if kind == "hang" and rank == int(who):
                    time.sleep(10_000)
                if kind == "die" and rank == int(who):
                    os._exit(7)
                dist.all_reduce(torch.ones(1))
                return None if rank else {"error": "dry"} if kind == "error" else {"dry_leg": "ok"}
            leg = sharded_leg_guarded(fake_leg, out if rank == 0 else None, rank, what="the dry stand-in of the sharded leg")
            if rank == 0:
                out.setdefault("other_modes", {})["one_frame_sharded_over_the_ranks"] = leg
            leg_failed = isinstance(leg, dict) and "error" in leg
        if not leg_failed:
            dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if leg_failed:
        os._exit(0)


def attach_sharded_leg(out, sharded_leg):
    """the sharded leg's figures under other_modes of the replicas' line (never `value`)"""
    om = out.setdefault("other_modes", {}) if out.get("other_modes") is not None else out.__setitem__("other_modes", {}) or out["other_modes"]
    if not isinstance(sharded_leg, dict) or "error" in sharded_leg:
        om["one_frame_sharded_over_the_ranks"] = sharded_leg if isinstance(sharded_leg, dict) else {"error": "the leg produced nothing"}
        return
    om["one_frame_sharded_over_the_ranks"] = {
        "value": sharded_leg["value"], "unit": sharded_leg["unit"], "ms_per_frame": sharded_leg["ms_per_step"], "scaling": "strong",
        "workload": sharded_leg["config"]["workload"], "events_per_rank": sharded_leg["config"]["events_per_rank"],
        "collective_ms": sharded_leg["collective_ms"], "kernels_us": sharded_leg["roofline"]["avg_launch_us"],
        "parity": sharded_leg["parity"],
        "merge": sharded_leg["config"]["merge"], "fell_back": sharded_leg["config"]["fell_back"],
        "frames_in_flight": sharded_leg["config"]["frames_in_flight"],
        "collectives_issued_by": sharded_leg["config"]["collectives_issued_by"], "comm_note": sharded_leg["config"]["comm_note"],
        "Mevents_per_s_via_torch_distributed": sharded_leg["config"]["Mevents_per_s_via_torch_distributed"],
        "Mevents_per_s_one_frame_at_a_time": sharded_leg["config"]["Mevents_per_s_one_frame_at_a_time"],
        "collective_bytes_per_frame_and_rank": sharded_leg["config"]["collective_bytes_per_frame_and_rank"],
        "note": "bench.py --sharded on the same ranks: C-10M, the event buffer split by index; merge = columns: every time column "
                "on one rank (all-gather of the shards' last events), plain u16 frames SUM-all-reduced; merge = all_reduce: MIN "
                "all-reduce of the extrema + MAX all-reduce of the packed-key frame; collective time (one frame at a time) "
                "listed separately; the headline `value` is frame-level weak scaling without any collective"}


def emit(out):
    """the ONE JSON line, last on stdout"""
    # RCCL prints its version banner through C stdio, which (redirected) is flushed at exit, i.e. AFTER Python's own
    # buffer: flush it now so that the JSON line is the LAST line on stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def leg_timeout_s():
    return float(os.environ.get("XM_BENCH_LEG_TIMEOUT_S", "300"))


def sharded_leg_guarded(run_leg, out, rank, what="the sharded leg (C-10M over the ranks)"):
    """Runs `run_leg()` on every rank under a LegGuard.  If it does not come back (a collective that never completes, a rank that
    died: SIGTERM from the launcher), rank 0 prints the replicas' line it already holds -- with the leg's error in it -- and
    every rank leaves through os._exit(0), without destroying communicators that may be stuck.  Returns the leg's result."""
    from benchmodes.guard import LegGuard

    def on_expire(reason):
        sys.stderr.write(f"[bench] rank {rank}: {reason}\n")
        if rank == 0 and out is not None:
            attach_sharded_leg(out, {"error": reason})
            emit(out)
    try:
        with LegGuard(leg_timeout_s(), on_expire, exit_code=0, name=what):
            return run_leg()
    except Exception as e:  # never lose the replicas' line to the extra leg
        sys.stderr.write("bench.py: sharded leg failed on rank %d: %r\n" % (rank, e))
        return {"error": repr(e)[:300]}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)  # (does not return)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(json.dumps({"error": f"WORLD_SIZE={world} but --gpus {args.gpus}: the launcher's --nproc-per-node and --gpus must agree"}),
                  flush=True)
        sys.exit(2)
    if os.environ.get("XM_BENCH_DRY") == "1":
        return dry_run(args, rank, world)
    if args.esl_stream_child:
        return esl_stream_child(args, local_rank)

    import torch

    if torch.cuda.device_count() <= local_rank:
        if rank == 0:
            print(json.dumps({"error": f"rank {rank} has no GPU (visible: {torch.cuda.device_count()})", "n_gpus_requested": args.gpus}), flush=True)
        sys.exit(2)
    dist = None
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1 or args.sharded or os.environ.get("XM_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist_mod
        from benchmodes.guard import LegGuard
        dist = dist_mod
        torch.cuda.set_device(local_rank)

        def no_rendezvous(reason):  # nothing is measured yet: an error line, a non-zero exit
            if rank == 0:
                emit({"error": "RCCL rendezvous / first all-reduce: " + reason, "n_gpus_requested": args.gpus})
        with LegGuard(leg_timeout_s(), no_rendezvous, exit_code=3, name="the RCCL rendezvous of the ranks"):
            if "MASTER_ADDR" not in os.environ:  # single process (--sharded on one GPU): file rendezvous, nothing to resolve
                import tempfile
                dist.init_process_group("nccl", init_method=f"file://{tempfile.mkdtemp()}/rdzv", rank=0, world_size=1,
                                        device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # the ranks count each other over RCCL before anything is measured
            one = torch.ones(1, device=dev)
            dist.all_reduce(one)
            ranks_seen = int(one.item())
        if ranks_seen != world:
            raise SystemExit(f"RCCL all-reduce saw {ranks_seen} ranks, WORLD_SIZE is {world}")
    else:
        torch.cuda.set_device(local_rank)
    if args.lib_option:
        from x_maps_amd import _native as xm_native
        for kv in args.lib_option:
            k, _, v = kv.partition("=")
            xm_native.debug_option(k, v)
    out = None
    leg_failed = False
    try:
        if args.sharded:
            out = bench_sharded(args, torch, dist, dev, rank, local_rank, world)
        elif args.esl:
            out = bench_esl(args, torch, dist, dev, rank, local_rank, world)
        elif args.graph:
            out = bench_graph(args, torch, dist, dev, rank, local_rank, world)
        else:
            out = bench_stream(args, torch, dist, dev, rank, local_rank, world)
            # N > 1: the frame-level replicas above need no collective.  The path's real exchange step -- one 10 M-event frame sharded
            # by event index over the ranks (BASELINE configs[3]) -- is measured behind them on the same ranks (never `value`): every
            # rank takes part, rank 0 keeps the figures.  Rank 0 holds the finished replicas' line by now: whatever happens inside
            # the leg, that line is printed (sharded_leg_guarded).
            if (world > 1 or os.environ.get("XM_BENCH_FORCE_SHARDED_LEG") == "1") and dist is not None and not args.no_other_modes:
                import copy  # (XM_BENCH_FORCE_SHARDED_LEG + XM_BENCH_FORCE_DIST: the tests exercise this leg on a one-GPU box)
                if rank == 0:
                    out["n_gpus"], out["rccl_ranks_seen"] = world, ranks_seen
                    sys.stderr.write("[bench] replicas measured: %s\n" % json.dumps({k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step")}))
                    sys.stderr.flush()
                a2 = copy.copy(args)
                a2.steps, a2.no_cpu_baseline, a2.single_block, a2.as_leg = 40, True, False, True
                leg = sharded_leg_guarded(lambda: bench_sharded(a2, torch, dist, dev, rank, local_rank, world), out, rank)
                leg_failed = isinstance(leg, dict) and "error" in leg
                if rank == 0 and (leg is not None):
                    attach_sharded_leg(out, leg)
    finally:
        if dist is not None and not leg_failed:  # (after a failed leg a communicator may be stuck: leave it alone)
            dist.destroy_process_group()
    if rank == 0 and out is not None:
        assert out["n_gpus"] == args.gpus == world, (out["n_gpus"], args.gpus, world)
        out["rccl_ranks_seen"] = ranks_seen if dist is not None else None
        emit(out)
    if leg_failed:
        sys.stdout.flush(), sys.stderr.flush()
        os._exit(0)


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
            measured, detail = measure_traffic(os.path.abspath(__file__), flags)
        finally:
            os.environ.pop("XM_BENCH_PMC_CHILD", None)
        apply_measured_traffic(roofline, measured, detail, s_frame, fps)
    elif roofline.get("traffic_source"):
        roofline["traffic_measured_in_run"] = False
        roofline["traffic_source"] += "; " + traffic_file_age()

    # ---- CPU baseline (rank 0 at N = 1 only) ---------------------------------------------------------------------------
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline_leg(args, O, tables, host_frames[0], n_ev, camera, bgr_out is not None)

    # ---- the same frames with other engine settings (extra information, never the headline `value`) -----------------------
    eng.close()  # one engine at a time: two engines would share the high-priority hardware queues
    other_modes = None
    if world == 1 and not args.no_other_modes:
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
        out["other_configs"] = other_config_legs(args, torch, dist, dev, local_rank)
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


# =====================================================================================================================
# --esl: configs[0] / configs[2] stand-in -- ESL-like frames (the recording itself is not available offline)
# =====================================================================================================================
def bench_esl(args, torch, dist, dev, rank, local_rank, world):
    from x_maps_amd import XMapsEngine
    from x_maps_amd import rig
    from x_maps_amd import synthetic as S
    from x_maps_amd.ingest import DeviceIngest

    camera = args.camera_perspective
    cp, tables, _, _ = rig.make_esl_like(row_stride=13, device=local_rank)
    B = args.batch  # frames per call (0: one frame per call, what DepthReprojectionPipe.process_ev_frame supplies)
    G = args.groups_in_flight if B else 1
    nf = max(8, B * G)
    host = [rig.render_events(cp, tables, row_stride=13, seed=rank * nf + f)[0] for f in range(nf)]
    lens = [len(e) for e in host]
    n_mean = float(np.mean(lens))
    slots = args.slots or (max(4, B * G) if B else 4)
    eng = XMapsEngine(tables, camera_perspective=camera, device=local_rank, n_slots=slots)
    info = eng.cols_info()
    H, W = eng.out_h, eng.out_w
    bgr_b = 0 if args.no_bgr else 3
    dev_frames = [torch.from_numpy(e.view(np.uint8).reshape(-1, 16).copy()).to(dev) for e in host]
    n_out = max(slots, 1)
    depth_out = torch.empty((n_out, H, W), dtype=torch.float32, device=dev)
    bgr_out = None if args.no_bgr else torch.empty((n_out, H, W, 3), dtype=torch.uint8, device=dev)
    groups = []
    if B:
        for g in range(G):
            fr = host[g * B:(g + 1) * B]
            offs = np.zeros(B + 1, np.uint64)
            offs[1:] = np.cumsum([len(e) for e in fr])
            rec = np.empty(int(offs[-1]), S.EVENT_CD_DTYPE)
            for i, e in enumerate(fr):
                rec[int(offs[i]):int(offs[i + 1])] = e
            aos = torch.from_numpy(rec.view(np.uint8).reshape(-1, 16).copy()).to(dev)
            soa = tuple(torch.from_numpy(np.ascontiguousarray(rec[k]).view(np.int16) if k != "t" else np.ascontiguousarray(rec[k])).to(dev)
                        for k in ("x", "y", "t"))
            groups.append((aos, offs, soa))
    torch.cuda.synchronize()
    parity, O = None, None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import xmaps_oracle as O

        def ref_of(e):
            return O.process_ev_frame(tables, e["x"].astype(np.int64), e["y"].astype(np.int64), np.ascontiguousarray(e["t"]),
                                      camera_perspective=camera, want_bgr=bgr_out is not None)
        d, b, st = eng.process_events(host[0], want_bgr=bgr_out is not None)
        ref = ref_of(host[0])
        parity = depth_parity(d, ref["depth"])
        if b is not None:
            parity["bgr_equal"] = bool(np.array_equal(b, ref["bgr"]))
        parity["n_inliers_equal"] = bool(st.n_inliers == int(ref["mask"].sum()))
        ok = parity["depth_max_rel_err"] <= 1e-4 and parity["empty_mask_equal"] and parity.get("bgr_equal", True)
        if B:  # the group path: first and last frame of group 0
            aos, offs, _ = groups[0]
            eng.process_events_batch_device(aos.data_ptr(), offs, depth_out[0].data_ptr(), None if bgr_out is None else bgr_out[0].data_ptr())
            eng.sync()
            parity["group_first_frame_depth_bit_exact"] = bool(np.array_equal(depth_out[0].cpu().numpy(), ref["depth"]))
            parity["group_last_frame_depth_bit_exact"] = bool(np.array_equal(depth_out[B - 1].cpu().numpy(), ref_of(host[B - 1])["depth"]))
            ok = ok and parity["group_first_frame_depth_bit_exact"] and parity["group_last_frame_depth_bit_exact"]
        if not ok and not args.no_parity:
            print(json.dumps({"error": "parity check failed", "parity": parity}))
            sys.exit(1)

    def step_single(i):
        f = dev_frames[i % nf]
        o = i % min(slots, 4)
        eng.process_events_device(f.data_ptr(), lens[i % nf], False, depth_out[o].data_ptr(),
                                  None if bgr_out is None else bgr_out[o].data_ptr())

    def step_group(i):
        aos, offs, _ = groups[i % G]
        o = (i % (slots // B)) * B
        eng.process_events_batch_device(aos.data_ptr(), offs, depth_out[o].data_ptr(), None if bgr_out is None else bgr_out[o].data_ptr())

    step = step_group if B else step_single
    fps = B or 1
    tm = Timer(torch, dist, dev, eng.sync)
    for i in range(args.warmup):
        step(i)
    tm.prewarm(step, PREWARM_S)
    roofline = alg = pt = wl = None
    if rank == 0:
        if B:
            def prof_group(i):
                _, offs, (sx, sy, st_) = groups[i % G]
                return eng.profile_batch_device(sx.data_ptr(), sy.data_ptr(), st_.data_ptr(), None, offs, depth_out[0].data_ptr(),
                                                None if bgr_out is None else bgr_out[0].data_ptr())
            group = (B, prof_group)
            frames_soa = None
        else:
            group = None
            frames_soa = []
            for e in host[:8]:  # (profile_frame_device takes one n: frames of their own length, one by one)
                frames_soa.append((torch.from_numpy(np.ascontiguousarray(e["x"]).view(np.int16)).to(dev),
                                   torch.from_numpy(np.ascontiguousarray(e["y"]).view(np.int16)).to(dev),
                                   torch.from_numpy(np.ascontiguousarray(e["t"])).to(dev)))
        if B:
            roofline, alg, pt, wl = roofline_of(eng, None, n_mean, (None, None), tables, camera, bgr_b, world, group, wl_suffix="_esl",
                                                cell_bytes=2 if info["mode"] != "none" else 8)
        else:
            n0 = lens[0]
            roofline, alg, pt, wl = roofline_of(eng, frames_soa[:1], n0, (depth_out[0].data_ptr(), None if bgr_out is None else bgr_out[0].data_ptr()),
                                                tables, camera, bgr_b, world, None, wl_suffix="_esl",
                                                cell_bytes=2 if info["mode"] != "none" else 8)
    est = tm.agree(tm.prewarm(step, 0.1))
    steps = args.steps
    R = 1 if args.single_block else int(min(200, max(3, round(TARGET_TIMED_S / max(steps * est, 1e-6)))))
    el, enq = tm.blocks(lambda: [step(i) for i in range(steps)], R)
    elapsed = float(np.median(el))
    ev_per_step = float(np.mean([sum(lens[(i % G) * B:(i % G) * B + B]) if B else lens[i % nf] for i in range(steps)]))
    value = ev_per_step * steps * world / elapsed / 1e6
    paths = eng.path_counts()
    if rank != 0:
        eng.close()
        return None
    s_frame = elapsed / (steps * fps)
    pipeline_fractions(roofline, alg, pt, wl, value, world, s_frame, fps, helper_runs=paths["cols"] > 0 or paths["general"] > 0)
    # ---- other ways in (never `value`) -------------------------------------------------------------------------------
    other = {}
    if B and not args.no_other_modes:
        tm1 = Timer(torch, None, dev, eng.sync)
        e1 = tm1.prewarm(step_single, PREWARM_S)
        k1 = max(1, steps * fps)
        el1, _ = tm1.blocks(lambda: [step_single(i) for i in range(k1)], int(min(200, max(3, round(0.2 / max(k1 * e1, 1e-6))))))
        dt1 = float(np.median(el1))
        other["one_frame_per_call_eager"] = {"value": round(float(np.mean(lens)) * k1 / dt1 / 1e6, 2), "unit": "Mevents/s",
                                             "us_per_frame": round(dt1 / k1 * 1e6, 2), "frames_in_flight": min(slots, 4),
                                             "note": "xm_process_frame_aos per frame, asynchronous (device-resident records), three launches per frame"}
        # the same calls on a handle with XM_FLAG_ADAPTIVE_BATCH: frames that arrive while the GPU is busy leave as one group
        eng.sync()
        with XMapsEngine(tables, camera_perspective=camera, device=local_rank, n_slots=slots, adaptive_batch=True) as eng_a:
            def step_adaptive(i):
                o = i % slots
                eng_a.process_events_device(dev_frames[i % nf].data_ptr(), lens[i % nf], False, depth_out[o].data_ptr(),
                                            None if bgr_out is None else bgr_out[o].data_ptr())
            tm2 = Timer(torch, None, dev, eng_a.sync)
            e2 = tm2.prewarm(step_adaptive, PREWARM_S)
            el2, _ = tm2.blocks(lambda: [step_adaptive(i) for i in range(k1)], int(min(200, max(3, round(0.2 / max(k1 * e2, 1e-6))))))
            dt2 = float(np.median(el2))
            pa = eng_a.path_counts()
            ok_a = True
            if O is not None:  # parity of the last frame the adaptive handle wrote
                j = (k1 - 1) % nf
                ok_a = bool(np.array_equal(depth_out[(k1 - 1) % slots].cpu().numpy(), ref_of(host[j])["depth"]))
        other["one_frame_per_call"] = {"value": round(float(np.mean(lens)) * k1 / dt2 / 1e6, 2), "unit": "Mevents/s",
                                       "us_per_frame": round(dt2 / k1 * 1e6, 2), "slots": slots, "k1_paths": pa,
                                       "last_frame_depth_bit_exact": ok_a,
                                       "note": "xm_process_frame_aos per frame on a handle with XM_FLAG_ADAPTIVE_BATCH (asynchronous, "
                                               "device-resident records)"}
    # what the pipe does per projector frame: one synchronous host call, EventCD records in, BGR frame out
    for i in range(20):
        eng.process_events(host[i % nf], want_depth=False, want_bgr=True)
    lat = []
    for i in range(200):
        c0 = time.perf_counter()
        eng.process_events(host[i % nf], want_depth=False, want_bgr=True)
        lat.append(time.perf_counter() - c0)
    lat = np.array(lat) * 1e3
    # a camera-like stream through the device-side ingest and through the processor, end to end
    ingest = None
    if not args.no_host_path and world == 1:
        try:
            ingest = esl_stream_legs(eng, cp, tables, int(n_mean), O, camera, local_rank)
        except Exception as e:  # never lose the line to the extra legs
            ingest = {"error": repr(e)[:300]}
        # ... and once more in a process that never imports torch -- the reference's own situation (Metavision + NumPy + OpenCV):
        # there the library runs on ROCm's HIP runtime instead of the older copy PyTorch ships and loads first in this process
        if "error" not in ingest and not getattr(args, "no_stream_child", False):
            try:
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--esl-stream-child"] + (["--camera-perspective"] if camera else []),
                                   capture_output=True, text=True, timeout=240)
                ch = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else {"error": (r.stderr or "no output")[-300:]}
            except Exception as e:
                ch = {"error": repr(e)[:300]}
            ingest["in_a_process_without_torch"] = ch
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        e0 = host[0]
        cpu = cpu_baseline_leg(args, O, tables, (e0["x"].copy(), e0["y"].copy(), np.ascontiguousarray(e0["t"])), len(e0), camera,
                               bgr_out is not None)
        cpu["ms_per_frame"] = round(len(e0) / cpu["value"] / 1e3, 3)
        cpu["reference_published_ms_per_frame"] = ("2.67 +- 0.31 (Numba on a Threadripper PRO 5955WX, real ESL frames: BASELINE.md section 1; other "
                                                   "hardware -- the port above is 3-17x slower than that and flatters any GPU / CPU ratio)")
    out = {
        "metric": "Mevents/s to depth frame, ESL-like frames (640x480 camera, 1080x1920 projector, ~150k ev/frame)",
        "value": round(value, 2), "unit": "Mevents/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int64+f64", "data": "synthetic",
        "config": {"workload": "C-ESL stand-in: frames rendered from a 3-D scene with the reference's real calibration geometry "
                               "(data/ESL_calib_hhi.yaml), rect 1760x1320, X-map 1320x1080, projector view 1080x1920; the ESL recording "
                               "itself is not available offline",
                   "events_per_frame_mean": round(n_mean), "frames_per_step": fps, "frames_in_flight": slots,
                   "inputs": "EventCD AoS resident in HBM", "frames_per_s": round(steps * fps * world / elapsed, 1),
                   "us_per_frame": round(s_frame * 1e6, 3),
                   "launch": (f"a step = one group of {B} frames through ONE call (xm_process_batch_aos), {G} groups in flight" if B else
                              "one frame per call (xm_process_frame_aos), asynchronous"),
                   "k1": {"none": "one thread per event, 64-bit atomic keys (the X-map is not injective and the rig did not "
                                  "qualify for the owner tiles)",
                          "cols": "column tiles", "own": "owner tiles (csrc/xmaps_k1own.hpp): no atomics, plain u16 frame"}[info["mode"]],
                   "k1_geometry": info, "k1_paths_frames": paths, "frames_redone_on_general_path": eng.sorted_fallbacks()},
        "per_frame_host_call_ms": {"p50": round(float(np.percentile(lat, 50)), 4), "p99": round(float(np.percentile(lat, 99)), 4),
                                   "definition": "DepthReprojectionPipe.process_ev_frame's work: one synchronous call, EventCD records in "
                                                 "pageable host memory -> BGR frame in host memory (H2D + kernels + D2H)",
                                   "reference_published_ms_per_frame": "2.67 +- 0.31 on a Threadripper PRO 5955WX, ESL static scenes "
                                                                      "(BASELINE.md section 1; other hardware, real data: context only)"},
        "timing": {"prewarm_s": PREWARM_S, "blocks": int(R), "block_s_median": round(elapsed, 6),
                   "block_s_min": round(float(el.min()), 6), "block_s_max": round(float(el.max()), 6)},
        "roofline": roofline, "other_modes": other or None,
        "ingest_path": (ingest or {}).get("ingest_path") if ingest and "error" not in ingest else ingest,
        "stream_legs": {k: v for k, v in (ingest or {}).items() if k != "ingest_path"} or None,
        "cpu_baseline": cpu, "parity": parity,
    }
    eng.close()
    return out


def esl_stream_child(args, device):
    """--esl's stream legs in a process of their own that never imports torch (bench.py --esl starts it): prints one JSON line"""
    assert "torch" not in sys.modules
    from x_maps_amd import XMapsEngine
    from x_maps_amd import rig
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xmaps_oracle as O
    camera = args.camera_perspective
    cp, tables, _, _ = rig.make_esl_like(row_stride=13, device=device)
    n_mean = float(np.mean([len(rig.render_events(cp, tables, row_stride=13, seed=f)[0]) for f in range(8)]))
    with XMapsEngine(tables, camera_perspective=camera, device=device, n_slots=4) as eng:
        legs = esl_stream_legs(eng, cp, tables, int(n_mean), O, camera, device)
    assert "torch" not in sys.modules
    keep = ("Mevents_per_s_end_to_end", "frames_per_s", "ms_per_cut_frame", "ms_per_shown_frame", "frames_cut", "frames_shown", "activity_filter",
            "same_frames_as_host_trigger_finder", "first_frame_equals_oracle", "host_us_per_push", "same_frames_as_host_path")
    out = {k: {q: v[q] for q in keep if q in v} for k, v in legs.items() if isinstance(v, dict) and k != "stream"}
    out["note"] = ("the same legs in a process without torch (NumPy + the library only, as in the reference's application): the library runs "
                   "on ROCm's own HIP runtime")
    print(json.dumps(out), flush=True)


def esl_stream_legs(eng, cp, tables, n_mean, O, camera, device, n_frames=48):
    """BASELINE config 3 stand-in, the way the reference runs it (depth_reprojection_pipe.py:110-119 -> trigger_finder.py:146-189):
    a camera-like ESL-like stream (10 % negative events, gap noise, 60 Hz frames) as quarter-period packets of raw EventCD records.
      device ingest   xm_ingest_push_pinned: filters, buffering, pause detection, frame cut and K0/K1/K2 on the device, frames into
                      the pinned result ring (BGR only = what the reference's frame_callback gets; + depth; as fresh arrays)
      processor       DepthReprojectionProcessor.process_events, host trigger finder + one fused call per cut frame (the reference's
                      structure), and the same processor with device_ingest=True"""
    from x_maps_amd import rig
    from x_maps_amd import synthetic as S
    from x_maps_amd.depth_reprojection_processor import DepthReprojectionProcessor, RuntimeParams
    from x_maps_amd.ingest import DeviceIngest
    from x_maps_amd.trigger_finder import RobustTriggerFinder
    stream, _ = rig.render_stream(cp, tables, n_frames=n_frames, row_stride=13, seed=9)
    pin = eng.host_empty((len(stream),), S.EVENT_CD_DTYPE)
    pin[:] = stream
    packet = int(1e6 / 60 / 4)
    cuts = np.searchsorted(pin["t"], np.arange(pin["t"][0], pin["t"][-1] + packet, packet))
    packets = [pin[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    # what the reference's own chain cuts out of these packets on the host: polarity filter -> activity filter (the checker's
    # sequential form of this build's rule, oracle/ingest_oracle.py; `False`: the stage left out, round 4's pipeline) ->
    # RobustTriggerFinder
    import ingest_oracle as IO
    want_by, want_frames_by, kept_by = {}, {}, {}
    for act_on in (True, False):
        want, want_frames = [], []

        def on_frame(e, want=want, want_frames=want_frames):
            want.append((int(e["t"][0]), int(e["t"][-1]), len(e)))
            if len(want_frames) < 1:
                want_frames.append(np.array(e))
        tf = RobustTriggerFinder(60, on_frame)
        act = IO.ActivityFilterC(640, 480, int(1e6 / 60))
        kept = 0
        for pk in packets:
            pos = pk[pk["p"] == 1]
            if act_on:
                pos = act.process(pos)
            kept += len(pos)
            tf.process_events(pos)
        want_by[act_on], want_frames_by[act_on], kept_by[act_on] = want, want_frames, kept
    want = want_by[True]
    out = {"stream": {"frames_rendered": n_frames, "events": int(len(stream)), "packets": len(packets), "packet_us": packet,
                      "frames_the_host_trigger_finder_cuts": len(want_by[True]), "frames_without_the_activity_filter": len(want_by[False]),
                      "events_behind_polarity_filter": kept_by[False], "events_behind_activity_filter": kept_by[True],
                      "note": "ESL-like stand-in (rig.render_stream: real calibration geometry, rendered scene, 10 % negative events, "
                              "gap noise); the reference's trigger finder loses lock on some frames by design -- the device cuts the "
                              "same ones.  Every leg runs the reference's chain polarity filter -> activity-noise filter -> trigger "
                              "finder (depth_reprojection_pipe.py:110-119) unless its name says filter_off"}}

    def run(want_depth, views, label, act_on=True):
        want, want_frames = want_by[act_on], want_frames_by[act_on]
        with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 18, expected_events_per_frame=n_mean,
                          result_ring=n_frames + 2, want_depth=want_depth, want_bgr=True, activity_filter=act_on) as ing:
            # warm-up = the whole stream once, untimed: first launches of every kernel, and one round of DMA through every pinned
            # buffer of the fresh result ring (the first copies into new pinned memory run at a third of the later rate under the
            # HIP runtime PyTorch bundles: a start-up cost of a ring that a live pipe allocates once)
            for pk in packets:
                ing.push_pinned(pk)
            ing.flush(), ing.reset(), ing.poll(copy=False)
            # three timed passes over the stream, the median one reported (under the HIP runtime PyTorch bundles a fresh ingest's
            # first passes run at anything between 0.45 and 1.0 of its settled rate; in a process without torch they do not)
            passes = []
            for rep in range(3):
                if rep:
                    ing.reset(), ing.poll(copy=False)
                hs0 = ing.host_stats()
                c0 = time.perf_counter()
                for pk in packets:
                    ing.push_pinned(pk)
                c1 = time.perf_counter()
                ing.flush()
                got = ing.poll(copy=not views)
                t_pass = time.perf_counter() - c0
                same_pass = [(f.t_first, f.t_last, f.n_events) for f in got] == want and not any(f.lost or f.overflow for f in got)
                passes.append((t_pass, c1 - c0, len(got), hs0, ing.host_stats(), same_pass))
            all_dt = [round(p[0] * 1e3, 3) for p in passes]
            dt, push_dt, n_got, hs0, hs, _ = sorted(passes, key=lambda p: p[0])[1]
            c1 = c0 + push_dt
            same_all = all(p[5] for p in passes)  # (every pass cut the reference's frames; `got` = the last pass: its views are intact)
            same = same_all
            ok = None
            if got and same and O is not None:
                e0 = want_frames[0]
                ref = O.process_ev_frame(tables, e0["x"].astype(np.int64), e0["y"].astype(np.int64), np.ascontiguousarray(e0["t"]),
                                         camera_perspective=camera, want_bgr=True)
                ok = bool(np.array_equal(got[0].bgr, ref["bgr"])) and (not want_depth or bool(np.array_equal(got[0].depth, ref["depth"])))
            n_push = hs["pushes"] - hs0["pushes"]
            out[label] = {"Mevents_per_s_end_to_end": round(len(stream) / dt / 1e6, 2), "frames_per_s": round(len(got) / dt, 1),
                          "ms_per_cut_frame": round(dt / max(len(got), 1) * 1e3, 4), "frames_cut": len(got), "activity_filter": bool(act_on),
                          "same_frames_as_host_trigger_finder": bool(same), "first_frame_equals_oracle": ok,
                          "host_us_per_push": round((hs["host_seconds_in_push"] - hs["seconds_waiting_for_the_gpu"] - hs0["host_seconds_in_push"]
                                                     + hs0["seconds_waiting_for_the_gpu"]) / max(n_push, 1) * 1e6, 2),
                          "host_us_per_push_incl_backpressure": round((hs["host_seconds_in_push"] - hs0["host_seconds_in_push"]) / max(n_push, 1) * 1e6, 2),
                          "push_loop_ms": round((c1 - c0) * 1e3, 3), "staging_waits": hs["staging_waits"] - hs0["staging_waits"],
                          "passes_ms": all_dt,
                          "outputs": ("BGR u8" + (" + depth f32" if want_depth else "")) + (", views into the pinned result ring" if views else ", fresh arrays (copied out of the ring)"),
                          "pcie_GBps_out": round(len(got) * eng.out_h * eng.out_w * (3 + (4 if want_depth else 0)) / dt / 1e9, 2)}
    run(False, True, "ingest_path")                      # what frame_callback gets in the reference: the BGR frame
    run(False, True, "ingest_path_filter_off", act_on=False)
    run(True, True, "ingest_path_depth_and_bgr")
    run(False, False, "ingest_path_fresh_arrays")
    # LIVE latency: the stream pushed at its own pace -- a packet becomes available at its last time stamp (speed 1 = the
    # camera's 60 Hz, 10 = ten times as fast) -- while the host polls; per frame: xm_ingest_push_pinned of the packet that cut it
    # called -> xm_ingest_poll hands the frame out (BGR view in the pinned ring).  The reference's loop is such a live system
    # (depth_reprojection.py:62-78; timing_watchdog.py:17-33 measures how far it falls behind).
    def run_paced(speed):
        with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 18, expected_events_per_frame=n_mean,
                          result_ring=n_frames + 2, want_depth=False, want_bgr=True, activity_filter=True) as ing:
            for pk in packets:
                ing.push_pinned(pk)
            ing.flush(), ing.reset(), ing.poll(copy=False)
            t_first = int(packets[0]["t"][0])
            lat, push_at, got_n = [], {}, 0
            base_push = ing.host_stats()["pushes"]
            c0 = time.perf_counter()

            def drain():
                nonlocal got_n
                for f in ing.poll(copy=False):
                    now = time.perf_counter()
                    got_n += 1
                    if f.push_seq - base_push in push_at and not f.lost:
                        lat.append(now - push_at[f.push_seq - base_push])
            for k, pk in enumerate(packets):
                due = c0 + (int(pk["t"][-1]) - t_first) / 1e6 / speed
                while time.perf_counter() < due:
                    drain()
                push_at[k + 1] = time.perf_counter()
                ing.push_pinned(pk)
            end = time.perf_counter() + 0.05
            while time.perf_counter() < end and got_n < len(want_by[True]):
                drain()
            ing.flush()
            drain()
        la = np.array(lat) * 1e3
        return {"speed": speed, "frames": int(len(la)), "frames_expected": len(want_by[True]), "each_ms": [round(float(v), 3) for v in la],
                "push_to_frame_visible_ms": {"p50": round(float(np.percentile(la, 50)), 4), "p99": round(float(np.percentile(la, 99)), 4),
                                             "max": round(float(la.max()), 4)} if len(la) else None}
    try:
        out["paced"] = {"real_time": run_paced(1.0), "ten_times": run_paced(10.0),
                        "definition": "ESL-like stream, quarter-period packets pushed when their last event's time has come (activity "
                                      "filter on, BGR views); latency = call of xm_ingest_push_pinned for the packet that completes "
                                      "a frame -> xm_ingest_poll returns that frame (H2D of the packet, ingest kernels, verdict, "
                                      "K0/K1/K2, 6.2 MB D2H, sequence number)"}
    except Exception as e:
        out["paced"] = {"error": repr(e)[:300]}
    try:  # the same stream as the recording stores it (EVT 3.0 words), one projector period per chunk, decoded on the device
        from x_maps_amd import evt3
        cuts3 = np.searchsorted(pin["t"], np.arange(pin["t"][0], pin["t"][-1] + 4 * packet, 4 * packet))
        chunks = []
        for a, b in zip(cuts3[:-1], cuts3[1:]):
            if b > a:
                w = evt3.encode_evt3_singles(pin[a:b])
                pw = eng.host_empty(w.shape, np.uint16)
                pw[:] = w
                chunks.append(pw)
        n_words = int(sum(len(c) for c in chunks))
        for act_on, label in ((True, "from_evt3_words_period_chunks"), (False, "from_evt3_words_period_chunks_filter_off")):
            with DeviceIngest(eng, 60, capacity_events=1 << 21, max_packet_events=1 << 19, expected_events_per_frame=n_mean,
                              result_ring=n_frames + 2, want_depth=False, activity_filter=act_on) as ing, \
                    evt3.DeviceEvt3Decoder(eng, max_words=max(len(c) for c in chunks)) as dec:
                for c in chunks:  # (warm-up: the whole stream once, see above)
                    dec.push(ing, c, pinned=True, count=False)
                ing.flush(), ing.reset(), ing.poll(copy=False), dec.reset()
                hs0 = ing.host_stats()
                c0 = time.perf_counter()
                for c in chunks:
                    dec.push(ing, c, pinned=True, count=False)
                ing.flush()
                got3 = ing.poll(copy=False)
                dt3 = time.perf_counter() - c0
                hs = ing.host_stats()
                over = max([f.overflow for f in got3] + [0])
                seq_pk = ing.activity_sequential_packets() if act_on else 0
            out[label] = {
                "Mevents_per_s_end_to_end": round(len(stream) / dt3 / 1e6, 2), "frames_cut": len(got3), "chunks": len(chunks), "overflow": over,
                "activity_filter": act_on, "chunks_judged_sequentially": seq_pk,
                "host_us_per_push": round((hs["host_seconds_in_push"] - hs["seconds_waiting_for_the_gpu"] - hs0["host_seconds_in_push"]
                                           + hs0["seconds_waiting_for_the_gpu"]) / max(hs["pushes"] - hs0["pushes"], 1) * 1e6, 2),
                "host_us_per_push_incl_backpressure": round((hs["host_seconds_in_push"] - hs0["host_seconds_in_push"]) / max(hs["pushes"] - hs0["pushes"], 1) * 1e6, 2),
                "bytes_per_event_over_pcie": round(2.0 * n_words / len(stream), 2), "processed_in_seconds": round(dt3, 4)}
    except Exception as e:
        out["from_evt3_words_period_chunks"] = {"error": repr(e)[:200]}

    # the reference's own structure: DepthReprojectionProcessor.process_events per packet (pageable packets, as Metavision hands them)
    def run_processor(device_ingest, views, label):
        shown = []

        class Window:
            def should_close(self):
                return False

            def show_async(self, img):
                shown.append((img.shape, int(img[::97, ::89].sum())))  # (consumes the frame inside the callback)
        params = RuntimeParams(camera_width=640, camera_height=480, projector_width=tables["proj_w"], projector_height=tables["proj_h"],
                               projector_fps=60, z_near=tables.get("z_near", 0.1), z_far=tables.get("z_far", 1.2), calib=None,
                               projector_time_map=None, no_frame_dropping=True, camera_perspective=camera, tables=tables, device=device,
                               device_ingest=device_ingest, ingest_frame_views=views, ingest_result_ring=64)
        pk_pageable = [np.array(pk) for pk in packets]
        with DepthReprojectionProcessor(params, window=Window()) as proc:
            for pk in pk_pageable:  # (warm-up: the whole stream once, see above)
                proc.process_events(pk)
            proc.flush(), proc.reset()
            shown.clear()
            c0 = time.perf_counter()
            for pk in pk_pageable:
                proc.process_events(pk)
            proc.flush()
            dt = time.perf_counter() - c0
        out[label] = {"Mevents_per_s_end_to_end": round(len(stream) / dt / 1e6, 2), "frames_per_s": round(len(shown) / dt, 1),
                      "ms_per_shown_frame": round(dt / max(len(shown), 1) * 1e3, 4), "frames_shown": len(shown),
                      "same_number_of_frames_as_host_trigger_finder": len(shown) == len(want)}
        return shown
    try:
        a = run_processor(False, False, "full_replay_through_processor_host_trigger_finder")
        b = run_processor(True, True, "full_replay_through_processor_device_ingest")
        out["full_replay_through_processor_device_ingest"]["same_frames_as_host_path"] = bool(a == b)
        out["full_replay_through_processor_host_trigger_finder"]["note"] = (
            "DepthReprojectionProcessor.process_events(packet): polarity filter (NumPy) + activity filter (one GPU call per packet: "
            "xm_activity_process) + RobustTriggerFinder in NumPy on the host, one "
            "synchronous fused call (H2D + K1 + K2 + D2H of the BGR frame) per cut frame: the reference's structure "
            "(reference_published_ms_per_frame 2.67 on a Threadripper PRO 5955WX for the frame stage alone)")
        out["full_replay_through_processor_device_ingest"]["note"] = (
            "the same calls with RuntimeParams(device_ingest=True, ingest_frame_views=True): packets are staged and pushed, frames are "
            "polled after every packet and handed to the window as views into the pinned result ring")
    except Exception as e:
        out["full_replay_through_processor"] = {"error": repr(e)[:300]}
    return out


# =====================================================================================================================
# --sharded: configs[3], C-10M frames sharded by event index over the ranks
# =====================================================================================================================
def bench_sharded(args, torch, dist, dev, rank, local_rank, world):
    from x_maps_amd import XMapsEngine
    from x_maps_amd import synthetic as S
    from x_maps_amd.sharded import GpuShardProvider, ShardedFrameProcessor, shard_bounds

    cfg = S.C_10M
    tables = S.make_tables(cfg)
    camera = args.camera_perspective
    n_ev = cfg.n_events
    nf = min(args.frames or 4, 4)
    a, b = shard_bounds(n_ev, rank, world)
    eng = XMapsEngine(tables, camera_perspective=camera, device=local_rank)
    K = max(1, min(args.lanes, nf))
    shards, host0, host_f = [], None, {}
    for f in range(nf):
        x, y, t, _ = S.to_soa(S.make_events(cfg, frame=f))
        if f == 0 and rank == 0:
            host0 = (x, y, t)
        if f < K and rank == 0:
            host_f[f] = (x, y, t)  # (lane k's first frame is frame k: each lane is checked against the oracle)
        shards.append(tuple(torch.from_numpy(v[a:b].copy()).to(dev) for v in (x.view(np.int16), y.view(np.int16), t)) + (None,))
    torch.cuda.synchronize()
    prov = GpuShardProvider(eng, dev)
    merge = args.merge
    if merge == "columns" and (camera or eng.shard_cols_info(n_ev) is None):
        merge = "all_reduce"  # (camera view / rigs whose X-map is not injective: the packed keys)
    state = {}

    def make(merge):
        for ln in state.get("lanes", [])[1:]:
            ln["eng"].close()
        lanes = []
        for k in range(K):  # lane k: frames k, k + K, ... on an engine of its own
            e = eng if k == 0 else XMapsEngine(tables, camera_perspective=camera, device=local_rank)
            pv = prov if k == 0 else GpuShardProvider(e, dev)
            pr = ShardedFrameProcessor(pv, dist, always_reduce=True, merge=merge)  # world 1: the collectives are issued all the same
            # the shards as a host keeps them resident for the columns path: headroom in front for the predecessor's last column
            res = {f: pr.columns_resident(shards[f], n_ev) for f in range(nf) if f % K == k} if merge == "columns" else None
            lanes.append(dict(eng=e, prov=pv, proc=pr, resident=res))
        state.update(lanes=lanes, merge=merge, proc=lanes[0]["proc"])

    make(merge)

    def process(i, want_bgr, lane=None):
        """frame i (of the nf resident ones) on its lane; lane = k: the i-th of lane k's own frames"""
        f = i % nf if lane is None else lane + K * (i % (nf // K))
        ln = state["lanes"][f % K]
        if state["merge"] == "columns":
            return ln["proc"].process_shard_columns(*ln["resident"][f], want_bgr=want_bgr)
        return ln["proc"].process_shard(shards[f], a, want_bgr=want_bgr)

    def sync():
        for ln in state["lanes"]:
            ln["eng"].sync()
        torch.cuda.synchronize()

    # Parity of frame 0 against the unsharded C oracle.  The verdict is COLLECTIVE (rank 0 checks, every rank hears): a rank
    # that left on its own would strand the others in the next collective.  A columns merge that fails falls back to the packed
    # keys and says so; a failure of those ends the leg on every rank together.
    refs = {}
    fell_back = None
    while True:
        merge = state["merge"]
        outs = [process(k, not args.no_bgr) for k in range(K)]  # every lane's first frame
        sync()
        cols_failed = any([ln["proc"].columns_failed() for ln in state["lanes"]]) if merge == "columns" else None  # (collectives: every rank)
        parity = None
        ok = True
        if rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from c_oracle import COracle
            for k, (depth, bgr) in enumerate(outs):
                if k not in refs:
                    r_ = COracle(tables, camera, omp=True).process_ev_frame(*host_f[k], want_events=False)
                    refs[k] = {q: (v.copy() if isinstance(v, np.ndarray) else v) for q, v in r_.items()}
                p = depth_parity(depth.cpu().numpy(), refs[k]["depth"])
                if bgr is not None:
                    p["bgr_equal"] = bool(np.array_equal(bgr.cpu().numpy(), refs[k]["bgr"]))
                if parity is None:
                    parity = p
                else:  # (the line shows the worst lane)
                    parity = {q: (max(parity[q], p[q]) if q == "depth_max_rel_err" else (parity[q] and p[q])) for q in parity}
            parity["checker"] = f"C/OpenMP oracle, unsharded frames 0..{K - 1} (one per lane)"
            if cols_failed is not None:
                parity["no_piece_objected"] = not cols_failed
            ok = bool(parity["depth_max_rel_err"] <= 1e-4 and parity.get("no_piece_objected", True) and parity["empty_mask_equal"]
                      and parity.get("bgr_equal", True)) or args.no_parity
            if merge == "columns" and os.environ.get("XM_BENCH_TEST_FAIL_COLUMNS") == "1":  # (the tests walk the fall-back)
                ok = False
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend() != "gloo" else "cpu")
        dist.broadcast(flag, src=0)
        if int(flag.item()):
            break
        if merge == "columns":
            fell_back = {"from": "columns", "parity_of_columns": parity}
            if rank == 0:
                print(f"[bench] sharded: the columns merge failed parity ({parity}); falling back to the packed keys", file=sys.stderr)
            make("all_reduce")
            continue
        if rank == 0:
            print(json.dumps({"error": "parity check failed", "parity": parity}))
        if getattr(args, "as_leg", False):
            raise RuntimeError("sharded leg: parity check failed")
        sys.exit(1)
    merge = state["merge"]
    proc = state["proc"]

    # collective time: torch events on the engine's stream around the collectives
    ev_pairs = []
    orig = {"_all_reduce": proc._all_reduce, "_reduce_scatter_max": proc._reduce_scatter_max, "_all_gather": proc._all_gather}

    def timed(fn):
        def wrapped(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()  # current stream = the engine's stream (process_shard runs under provider.collective_stream())
            fn(*a)
            e1.record()
            ev_pairs.append((e0, e1))
        return wrapped

    tm = Timer(torch, dist, dev, sync)

    # The library's own communicators (xm_shard_comm_*), one per lane: the timed loop runs on them when every rank could set
    # them up and their frames pass the same check; the torch.distributed lanes above stay for the per-kernel / per-collective
    # timing pass and as the fall-back.  Every decision here is collective (an all-reduce / broadcast of the verdict).
    comms, comm_note = None, None
    on = dev if dist.get_backend() != "gloo" else "cpu"
    if args.comm == "library" and merge in ("columns", "all_reduce"):
        from x_maps_amd.sharded import ShardComm
        try:
            ShardComm.new_id()  # (local probe: is librccl there with the entry points?)
            can = 1
        except Exception as e:
            can, comm_note = 0, repr(e)[:200]
        flag = torch.tensor([can], dtype=torch.int32, device=on)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            comms = [ShardComm.over_torch_dist(ln["eng"], dist, n_ev, dev) for ln in state["lanes"]]

            def process_lib(i, want_bgr):
                f = i % nf
                if merge == "columns":
                    return comms[f % K].frame(*state["lanes"][f % K]["resident"][f], want_bgr=want_bgr)
                return comms[f % K].frame_keys(shards[f], a, want_bgr=want_bgr)

            outs = [process_lib(k, not args.no_bgr) for k in range(K)]
            sync()
            bad = any([c.failed() for c in comms]) if merge == "columns" else False  # (collectives)
            ok = True
            if rank == 0:
                for k, (depth, bgr) in enumerate(outs):
                    ok = ok and np.array_equal(depth.cpu().numpy(), refs[k]["depth"]) and (bgr is None or np.array_equal(bgr.cpu().numpy(), refs[k]["bgr"]))
                ok = bool(ok and not bad) or args.no_parity
                parity["library_communicator_frames_equal_oracle"] = bool(ok)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=on)
            dist.broadcast(flag, src=0)
            if not int(flag.item()):
                comm_note = "frames through the library's communicator differed from the oracle: torch.distributed path timed instead"
                if rank == 0:
                    print("[bench] sharded: " + comm_note, file=sys.stderr)
                for c in comms:
                    c.close()
                comms = None
        else:
            comm_note = comm_note or "another rank could not set the library's communicator up"

    def step(i):
        if comms is not None:
            process_lib(i, not args.no_bgr)
        else:
            process(i, not args.no_bgr)

    for i in range(min(args.warmup, 50)):
        step(i)
    est = tm.agree(tm.prewarm(step, PREWARM_S))
    steps = args.steps if args.steps != 2000 else 200  # default K for this workload: 200 frames of 10 M events
    R = 1 if args.single_block else int(min(50, max(3, round(TARGET_TIMED_S / max(steps * est, 1e-6)))))
    el, enq = tm.blocks(lambda: [step(i) for i in range(steps)], R)
    elapsed = float(np.median(el))
    value = float(n_ev) * steps / elapsed / 1e6  # the frame is shared by all ranks: strong scaling
    one_lane, via_torch, enq_torch = None, None, None
    if comms is not None:  # the same lanes with Python / torch.distributed issuing the collectives (x_maps_amd.sharded)
        elt, enqt = tm.blocks(lambda: [process(i, not args.no_bgr) for i in range(steps)], max(3, R // 2))
        via_torch, enq_torch = float(n_ev) * steps / float(np.median(elt)) / 1e6, float(np.median(enqt)) / steps * 1e6
    if K > 1:  # the same frames one at a time (lane 0 alone, torch.distributed): what rounds 1-3 measured
        el1, _ = tm.blocks(lambda: [process(i, not args.no_bgr, lane=0) for i in range(steps)], max(3, R // 2))
        one_lane = float(n_ev) * steps / float(np.median(el1)) / 1e6
    # collective time, measured in a separate short pass (event records between the enqueues cost host time)
    for k, f in orig.items():
        setattr(proc, k, timed(f))
    # ... and the shard's three kernels the same way (K0 extrema of the shard, K1 scatter with global event indices, K2 on the
    # merged key frame): torch events on the engine's stream, which is torch's current stream inside process_shard
    k_pairs = {"minmax_into": [], "scatter": [], "finish": [], "finish_u16": [], "finish_u16_band": [], "cols_pack": [], "cols_scatter": [], "cols_finish": []}
    p_orig = {k: getattr(prov, k) for k in k_pairs}

    def timed_k(name, fn):
        def wrapped(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            k_pairs[name].append((e0, e1))
            return r
        return wrapped
    for k, f in p_orig.items():
        setattr(prov, k, timed_k(k, f))
    k1_alone = []
    if merge == "columns":
        from x_maps_amd import _native as xm_native
        xm_native.debug_option("XM_SHARD_PROFILE", "1")
    for i in range(20):  # (lane 0 alone: a frame at a time, so that an event pair brackets one kernel chain and nothing else)
        process(i, not args.no_bgr, lane=0)
        if merge == "columns":
            k1_alone.append(eng.shard_cols_last_k1_ms())
    sync()
    if merge == "columns":
        xm_native.debug_option("XM_SHARD_PROFILE", None)
    for k, f in orig.items():
        setattr(proc, k, f)
    for k, f in p_orig.items():
        setattr(prov, k, f)
    med = lambda k: float(np.median([e0.elapsed_time(e1) for e0, e1 in k_pairs[k]])) if k_pairs[k] else 0.0
    if merge == "columns":  # (the pack of the shard's last events in the helper slot; K1 = prepare + boundary pass + column tiles)
        k_ms = [med("cols_pack"), med("cols_scatter"), med("cols_finish")]
    else:
        k_ms = [med(k) for k in ("minmax_into", "scatter", {"all_reduce": "finish", "reduce_scatter": "finish_u16", "bands": "finish_u16_band"}[merge])]
    # extrema + key frame | extrema + reduce-scatter + all-gather | extrema + reduce-scatter + depth + BGR (the halo exchange is
    # point to point and not timed here) | headers + last events + u16 frame
    per_frame = {"all_reduce": 2, "reduce_scatter": 3, "bands": 3 if args.no_bgr else 4, "columns": 2}[merge]
    coll = np.array([e0.elapsed_time(e1) for e0, e1 in ev_pairs]).reshape(-1, per_frame)
    coll_ms = torch.tensor([float(np.median(coll[:, 0])), float(np.median(coll[:, 1:].sum(axis=1)))], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(coll_ms, op=dist.ReduceOp.MAX)
    for c in comms or []:
        c.close()
    if rank != 0:
        for ln in state["lanes"]:
            ln["eng"].close()
        return None
    kshape = eng.key_shape
    roofline, alg, pt = roofline_dict(np.array(k_ms + [elapsed / steps * 1e3]), None, b - a, 1, tables, camera, 0 if args.no_bgr else 3,
                                      ("camera" if camera else "projector") + ("_sharded" if merge == "columns" else "_sharded_keys"),
                                      "torch.cuda.Event pairs recorded on the engine's stream (torch's current stream inside "
                                      "process_shard) around the shard's three kernel launches, 20 frames, median; k_minmax = the "
                                      "shard's extrema pass K0; k_scatter processes THIS rank's events (events_per_rank); k_frame "
                                      "runs on the merged key frame on every rank; merge = columns: k_minmax = the pack of the shard's last events, "
                                      "k_scatter = prepare (extrema, own / predecessor's last column) + boundary pass + column-tile K1", cell_bytes=8 if merge == "all_reduce" else 2)
    pipeline_fractions(roofline, alg, pt, ("camera" if camera else "projector") + ("_sharded" if merge == "columns" else "_sharded_keys"), value, 1, elapsed / steps, 1)
    if merge == "columns":  # what the k_scatter figures bracket here, and K1 alone beside them
        roofline["kernels"]["k_scatter"]["brackets"] = "k_shard_cols_prepare + k_cols_bounds_batch + k_scatter_cols_batch (three launches, one event pair)"
        roofline["k_scatter_brackets"] = roofline["kernels"]["k_scatter"]["brackets"]
        if k1_alone:
            k1_us = float(np.median(k1_alone[5:])) * 1e3
            a1 = 24.0 * (b - a)
            roofline["kernels"]["k_scatter_cols_batch_alone"] = {
                "avg_launch_us": round(k1_us, 2), "algorithmic_bytes_per_launch": a1, "frac_algorithmic": round(a1 / (k1_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                "timing": "HIP events tied to the K1 dispatch alone (hipExtLaunchKernelGGL; xm_shard_cols_last_k1_ms), 15 frames, median"}
            roofline["frac_k1_alone"] = roofline["kernels"]["k_scatter_cols_batch_alone"]["frac_algorithmic"]

    roofline["event_stream_read_roofline_frac_note"] = "whole frame (all ranks' events) per step time against ONE GPU's HBM read peak"
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        try:
            from c_oracle import COracle
            co = COracle(tables, camera, omp=True)
            c0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - c0 < args.cpu_seconds and reps < 20:
                co.process_ev_frame(*host0, want_events=False)
                reps += 1
            cpu = {"value": round(reps * n_ev / (time.perf_counter() - c0) / 1e6, 2), "unit": "Mevents/s", "cores": co.threads,
                   "kind": "port", "sample": f"{reps} x C-10M frame 0, fused C + OpenMP port (the 1-core NumPy port needs ~0.3 s/frame)"}
        except Exception as e:
            cpu = {"error": str(e)[:200]}
    out = {
        "metric": "Mevents/s to depth frame, 1280x720, 10M ev/frame, sharded by event index", "value": round(value, 2),
        "unit": "Mevents/s", "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
        "config": {"workload": f"C-10M: synthetic 10M events/frame, 1280x720 cam/proj, rect 3520x1980, event buffer sharded by "
                               f"index over {world} rank(s), " + ("every time column on one rank, u16 frames SUM-all-reduced over RCCL" if merge == "columns"
                                                                    else "packed-key frame MAX-all-reduced over RCCL") +
                               (" (camera view)" if camera else " (projector view)"),
                   "events_per_frame": n_ev, "events_per_rank": b - a, "key_frame_MB": round(kshape[0] * kshape[1] * 8 / 1e6, 1),
                   "host_synchronisations_per_frame": 0,
                   "merge": merge, "fell_back": fell_back, "frames_in_flight": K,
                   "collectives_issued_by": "the library (xm_shard_comm_frame: one native call per frame, an RCCL communicator per lane)"
                   if comms is not None else "torch.distributed (x_maps_amd.sharded.ShardedFrameProcessor)", "comm_note": comm_note,
                   "Mevents_per_s_via_torch_distributed": None if via_torch is None else round(via_torch, 1),
                   "host_enqueue_us_per_frame_via_torch_distributed": None if enq_torch is None else round(enq_torch, 2),
                   "Mevents_per_s_one_frame_at_a_time": None if one_lane is None else round(one_lane, 1), "collective_bytes_per_frame_and_rank": getattr(proc, "collective_bytes_per_frame", None),
                   "collectives_per_frame": (["all_gather of {first / last stamp, the shard's last events} (carries the extrema and every last column)",
                                              "all_reduce SUM uint32[u16 frame / 2] (disjoint cells)"] if merge == "columns" else
                                             ["all_reduce MIN int64[2] (frame extrema)"]) +
                                            ([] if merge == "columns" else
                                             ["all_reduce MAX int64[key frame]"] if merge == "all_reduce" else
                                             ["reduce_scatter MAX int64[key frame]", "all_gather u16[key frame] (decoded disparities)"]
                                             if merge == "reduce_scatter" else
                                             ["reduce_scatter MAX int64[key frame]", "send / recv of the band's halos (neighbours)",
                                              "all_reduce MAX of the partial projector frames (depth as int32, BGR u8)"])},
        "collective_ms": {("last_events_all_gather" if merge == "columns" else "extrema_min_all_reduce"): round(float(coll_ms[0]), 4),
                          "key_frame_merge": round(float(coll_ms[1]), 4),
                          "note": "median over 20 frames, torch events on the engine's stream around each all-reduce, max over ranks; "
                                  "with one rank RCCL still runs its kernels (always_reduce) but nothing crosses xGMI"},
        "timing": {"prewarm_s": PREWARM_S, "blocks": int(R), "block_s_median": round(elapsed, 6),
                   "block_s_min": round(float(el.min()), 6), "block_s_max": round(float(el.max()), 6),
                   "host_enqueue_us_per_frame": round(float(np.median(enq)) / steps * 1e6, 2)},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
    }
    for ln in state["lanes"]:
        ln["eng"].close()
    return out


if __name__ == "__main__":
    main()
