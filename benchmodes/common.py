"""What bench.py's modes share: constants, the timing protocol (Timer), parity / roofline helpers, the CPU baseline leg.
(The driver's contract -- flags, the ONE JSON line, the launch logic -- lives in ../bench.py.)"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s
PREWARM_S = float(os.environ.get("XM_BENCH_PREWARM_S", "0.35"))  # wall-clock pre-warm before anything is measured
TARGET_TIMED_S = 0.30  # the R timed blocks together


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
        # In a CHILD process that never imports torch, started with the OpenMP placement in its environment: inside this process
        # PyTorch's bundled OpenMP runtime is already up (its threads, its settings -- OMP_PROC_BIND set afterwards is not read),
        # and the leg swung between 16 and 100 Mev/s from run to run.  The MEDIAN frame is reported (a 128-thread barrier waits
        # for its slowest thread: one descheduled thread is one slow frame), mean and best beside it.
        import json as _json
        import subprocess
        import sys
        import tempfile
        keys = ("cam_mapx_i16", "cam_mapy_i16", "proj_x_map", "disp_proj_mapxy_i16")
        scal = {k: tables[k] for k in ("rect_w", "rect_h", "p03", "z_near", "z_far")}
        scal["x_offset"] = tables.get("x_offset", 4242)
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "frame.npz")
            np.savez(f, x=x, y=y, t=t, **{k: np.asarray(tables[k]) for k in keys}, scal=np.array(_json.dumps({k: float(v) for k, v in scal.items()})))
            code = ("import sys, json, time, numpy as np; sys.path.insert(0, %r); from c_oracle import COracle\n"
                    "d = np.load(%r); sc = json.loads(str(d['scal'])); tb = {k: d[k] for k in %r}\n"
                    "tb.update({k: (int(v) if k in ('rect_w', 'rect_h', 'x_offset') else v) for k, v in sc.items()})\n"
                    "co = COracle(tb, %r, omp=True, reuse_outputs=True)\n"
                    "x, y, t = d['x'], d['y'], d['t']\n"
                    "for _ in range(3): co.process_ev_frame(x, y, t, want_events=False)\n"
                    "ts = []; c0 = time.perf_counter()\n"
                    "while time.perf_counter() - c0 < %r and len(ts) < 200:\n"
                    "    a = time.perf_counter(); co.process_ev_frame(x, y, t, want_events=False); ts.append(time.perf_counter() - a)\n"
                    "print(json.dumps({'threads': co.threads, 'ts': ts}))\n") % (os.path.join(ROOT, "oracle"), f, keys, bool(camera), float(min(3.0, args.cpu_seconds)))
            env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_WAIT_POLICY="active")  # (active: the frame is seven parallel regions of ~0.1 ms)
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-300:])
        res = _json.loads(r.stdout.strip().splitlines()[-1])
        ts = sorted(res["ts"])
        med, mean, best_c = ts[len(ts) // 2], sum(ts) / len(ts), ts[0]
        cpu["all_cores_c_openmp"] = {"value": round(n_ev / med / 1e6, 2), "unit": "Mevents/s", "cores": res["threads"], "kind": "port",
                                     "mean": round(n_ev / mean / 1e6, 2), "best": round(n_ev / best_c / 1e6, 2),
                                     "p90_frame": round(n_ev / ts[min(len(ts) - 1, (len(ts) * 9) // 10)] / 1e6, 2),
                                     "slowest_frame_ms": round(ts[-1] * 1e3, 2),
                                     "sample": f"{len(ts)} x frame 0, after 3 warm-up frames, the MEDIAN frame; one thread per physical core "
                                               f"(OMP_PROC_BIND=close, OMP_PLACES=cores, OMP_WAIT_POLICY=active), outputs and scratch reused; in a child "
                                               f"process without PyTorch's OpenMP runtime (inside this one the same loop ran at 16-100 Mev/s: round 6); "
                                               f"`mean` includes the frames of the run that stalled (slowest_frame_ms)"}
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


_LEG_T0 = [None]


def leg_clock(name):
    """XM_BENCH_LEG_TIMES=1: wall time since the previous mark, to stderr (where a bench.py run spends its minutes)"""
    import os, sys, time
    if os.environ.get("XM_BENCH_LEG_TIMES") != "1":
        return
    now = time.perf_counter()
    if _LEG_T0[0] is not None:
        sys.stderr.write("[bench legs] %-44s %7.2f s\n" % (name, now - _LEG_T0[0]))
        sys.stderr.flush()
    _LEG_T0[0] = now
