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

# The modes live under benchmodes/ (stream = the default workload, graph, esl, sharded; common = the timing protocol, roofline and
# parity helpers; other = the compact legs of the other configs inside the default line; guard / pmc = the helpers of round 5):
# this file is the driver's contract -- the flags, the launch logic, the ONE JSON line.


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
    ap.add_argument("--groups-in-flight", type=int, default=None,
                    help="with --batch: slots = groups x B.  Default 4 (one group per stream / hardware queue of the handle: 149-151 -> "
                         "162-166 Gev/s against 3 in alternating runs, 5 / 6 / 8 no different); --esl: 2 (the ESL-like step is two "
                         "thirds K2, which is bound by what it writes: 0.256-0.259 ms per step against 0.275 with 4 in the same run)")
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
    args = ap.parse_args()
    if args.groups_in_flight is None:
        args.groups_in_flight = 2 if args.esl else 4
    return args


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
    from benchmodes.esl import bench_esl, esl_stream_child
    from benchmodes.graph import bench_graph
    from benchmodes.sharded import bench_sharded
    from benchmodes.stream import bench_stream
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


if __name__ == "__main__":
    main()
