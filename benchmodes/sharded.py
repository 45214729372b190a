"""bench.py --sharded: BASELINE configs[3], C-10M frames sharded by event index over the ranks (RCCL)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

from .common import (BENCH_PY, HBM_PEAK_GBS, PREWARM_S, ROOT, TARGET_TIMED_S, Timer, apply_measured_traffic, cpu_baseline_leg, depth_parity,
                     n_blocks_for, parity_ok, pipeline_fractions, roofline_dict, roofline_of, traffic_file_age)


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
            comms = []
            for ln in state["lanes"]:  # one communicator at a time (each creation is a collective), a barrier between them
                comms.append(ShardComm.over_torch_dist(ln["eng"], dist, n_ev, dev))
                dist.barrier()

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
    for i in range(20):  # (lane 0 alone: a frame at a time, so that an event pair brackets one kernel chain and nothing else)
        process(i, not args.no_bgr, lane=0)
    sync()
    k1_alone = []
    if merge == "columns":  # ... and K1 ALONE in a pass of its own (its dispatch carries HIP events there, which stretches the bracket)
        from x_maps_amd import _native as xm_native
        xm_native.debug_option("XM_SHARD_PROFILE", "1")
        ev_keep, kp_keep = list(ev_pairs), {k: list(v) for k, v in k_pairs.items()}
        for i in range(15):
            process(i, not args.no_bgr, lane=0)
            k1_alone.append(eng.shard_cols_last_k1_ms())
        sync()
        xm_native.debug_option("XM_SHARD_PROFILE", None)
        ev_pairs[:] = ev_keep  # (the brackets of this pass do not count)
        for k in k_pairs:
            k_pairs[k][:] = kp_keep[k]
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
                "timing": "HIP events tied to the K1 dispatch alone (hipExtLaunchKernelGGL; xm_shard_cols_last_k1_ms), a pass of 15 frames of its own, median of the last 10"}
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
