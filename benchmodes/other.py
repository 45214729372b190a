"""The other BASELINE configs as compact legs inside the default line (the driver runs only that one)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

from .common import (BENCH_PY, HBM_PEAK_GBS, PREWARM_S, ROOT, TARGET_TIMED_S, Timer, apply_measured_traffic, cpu_baseline_leg, depth_parity,
                     n_blocks_for, parity_ok, pipeline_fractions, roofline_dict, roofline_of, traffic_file_age)

from .esl import bench_esl
from .graph import bench_graph
from .sharded import bench_sharded

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
        om = out.get("other_modes") or {}
        if isinstance(om.get("groups_bgr_only"), dict):  # (configs 0 / 2: the output the reference's frame_callback gets)
            leg["groups_bgr_only"] = {k: v for k, v in om["groups_bgr_only"].items() if k != "note"}
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
            for k in ("full_replay_through_processor_default_params", "full_replay_through_processor_host_trigger_finder",
                      "full_replay_through_processor_device_ingest", "ingest_path_fresh_arrays", "ingest_path_depth_and_bgr"):
                if isinstance(sl.get(k), dict):
                    leg[k] = {a: b for a, b in sl[k].items() if a != "note"}
            ch = sl.get("in_a_process_without_torch")
            if isinstance(ch, dict):  # (the ingest leg and the processor's device-ingest leg: the two the reference's application runs)
                leg["in_a_process_without_torch"] = {k: ch[k] for k in ("ingest_path", "ingest_path_fresh_arrays", "full_replay_through_processor_default_params",
                                                                          "full_replay_through_processor_device_ingest", "error") if k in ch}
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
        if name == "esl":
            a.groups_in_flight = 2  # (as `bench.py --esl`: bench.py's --groups-in-flight)
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
