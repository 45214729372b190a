#!/usr/bin/env python3
"""Turn the rocprofv3 databases of tools/r05/make_profiles.sh into the summaries that get committed under profiles/:
<tag>_kernel_trace.md, <tag>_pmc.md and the traffic entries of pmc_traffic.json that bench.py reads (projector_groups,
camera_groups, projector_groups_esl, projector_sharded, projector).  On the GPU box everything goes to gpurun_out/<tag>_profiles/."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
rnd = sys.argv[2] if len(sys.argv) > 2 else "5"  # (tools/r06/make_profiles.sh passes 6)
src = os.path.join(ROOT, "gpurun_out", tag)
out = os.path.join(ROOT, "gpurun_out", tag + "_profiles")
os.makedirs(out, exist_ok=True)
SKIP = ("reset", "build_dlut", "k2_tables", "pix_to_u16", "cols_check", "build_x_map", "rocclr", "at::native", "elementwise", "Rccl", "rccl", "ncclDev")


def short(n):
    return n.split("(")[0].replace("void ", "")


def logical(n):
    if "k_minmax" in n or "k_cols_bounds" in n:
        return "k_minmax"
    if "k_scatter" in n:
        return "k_scatter"
    if "k_frame" in n:
        return "k_frame"
    return None


def trace_table(db):
    agg = defaultdict(list)
    for name, d in sqlite3.connect(db).execute("select name, end - start from kernels"):
        agg[short(name)].append(d)
    total = sum(sum(v) for v in agg.values())
    lines = ["| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"]
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / total < 0.002:
            continue
        lines.append(f"| `{name}` | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | {sum(v)/1e6:.3f} | {100*sum(v)/total:.1f} |")
    return "\n".join(lines), {k: sum(v) / len(v) / 1e3 for k, v in agg.items()}


def pmc_rows(pattern):
    rows = defaultdict(dict)
    for db in sorted(glob.glob(os.path.join(src, pattern))):
        try:
            for k, cn, v in sqlite3.connect(db).execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1, 2"):
                if "xm::" in k and not any(s in k for s in SKIP):
                    rows[short(k)][cn] = v
        except sqlite3.Error:
            pass
    return rows


SETS = (("groups", "C-1M (BASELINE configs[1]), projector view, groups of 32 frames, ONE group at a time", "python bench.py --groups-in-flight 1", 32, "projector_groups"),
        ("esl", "ESL-like frames (configs[0]/[2] stand-in), groups of 32, one group at a time", "python bench.py --esl --groups-in-flight 1", 32, "projector_groups_esl"),
        ("camg", "C-1M camera view, groups of 32, one group at a time", "python bench.py --groups-in-flight 1 --camera-perspective", 32, "camera_groups"),
        ("sharded", "C-10M (configs[3]) on one rank, merge = columns: pack / prepare + boundary pass + column-tile K1 + K2, RCCL all-gather + SUM all-reduce", "python bench.py --sharded --lanes 1 --comm torch", 1, "projector_sharded"),
        ("shardedkeys", "C-10M on one rank, merge = all_reduce (packed 64-bit keys: K0 + event-tile K1 with atomics + K2, two RCCL all-reduces)", "python bench.py --sharded --merge all_reduce --lanes 1 --comm torch", 1, "projector_sharded_keys"),
        ("single", "C-1M, one frame per call, one slot (single-frame launches)", "python bench.py --batch 0 --slots 1", 1, "projector"))
Q = "--no-cpu-baseline --no-other-modes --no-host-path --no-pmc"
avg_us = {}
with open(os.path.join(out, f"{tag}_kernel_trace.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of bench.py (MI355X, ROCm 7.2), round {rnd}\n\n"
            "Group modes: a bench step = one group of 32 frames through one xm_process_batch call = ONE launch each of the boundary pass\n"
            "(k_cols_bounds_batch), K1 (k_scatter_cols_batch: column tiles; k_scatter_own_batch: owner tiles) and K2 (k_frame_proj_pipe:\n"
            "persistent, software-pipelined blocks).  Per-frame cost = avg us / 32.\n\n")
    for key, title, cmd, fpl, wl in SETS + (("groups3", "the default bench command (4 groups in flight: launches of different groups overlap)", "python bench.py", 32, None),
                                            ("sharded2", "C-10M on one rank as bench.py --sharded runs it by default: 2 frames in flight (lanes), the library issuing the RCCL collectives (xm_shard_comm_frame) -- one frame's small kernels run beside the other's K1, so their durations stretch", "python bench.py --sharded", 1, None),
                                            ("graph", "60 frames x 1 M events from one captured hipGraph (configs[4])", "python bench.py --graph", 60, None),
                                            ("evt3", "the EVT 3.0 decoder alone: 20 chunks of 2 M events = 4 M words each (tools/evt3_probe.py)", None, 1, None),
                                            ("ingest", "the default bench's host / ingest legs (C-1M camera-like stream: records and EVT 3.0 words through the device ingest)", "python bench.py --no-cpu-baseline --no-other-modes --no-other-configs", 1, None),
                                            ("eslstream", "the ESL-like camera stream through the device ingest, ACTIVITY FILTER ON (the default), BGR views (tools/r05/act_probe.py 1 3: 4 x 191 packets / 43 cut frames)", None, 1, None),
                                            ("eslstream_off", "the same stream with the activity filter off (tools/r05/act_probe.py 0 3)", None, 1, None),
                                            ("b1", "a group of ONE C-1M frame through the column tiles (what a lone frame would cost there: boundary pass + K1 + K2, against 11.9 + 7.6 us on the compact-key path it takes today)", "python bench.py --batch 1 --groups-in-flight 1 --steps 200 --warmup 20 --no-other-configs", 1, None)):
        db = os.path.join(src, f"trace_{key}_results.db")
        if os.path.exists(db):
            tab, avg = trace_table(db)
            avg_us[key] = avg
            line = ("python tools/evt3_probe.py 2000000 20" if key == "evt3" else "python tools/r05/act_probe.py 1 3" if key == "eslstream" else
                    "python tools/r05/act_probe.py 0 3" if key == "eslstream_off" else cmd + " --no-pmc" if key == "ingest" else f"{cmd} {Q}")
            f.write(f"## {title}\n\n`rocprofv3 --kernel-trace --stats -- {line}`\n\n{tab}\n\n")
    for j in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
        if os.path.getsize(j):
            f.write(f"## {os.path.basename(j)} (un-profiled run on the same box)\n\n```json\n{open(j).read().strip().splitlines()[-1]}\n```\n\n")

traffic = {}
with open(os.path.join(out, f"{tag}_pmc.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 PMC counters per kernel (averages per dispatch; one --pmc group per run, --kernel-trace only)\n\n"
            "FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read\n"
            "(MI355X_MICROARCH.md, HBM section), so HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE.\n\n")
    for key, title, cmd, fpl, wl in SETS:
        rows = pmc_rows(f"pmc_{key}_*_results.db")
        if not rows:
            continue
        f.write(f"## {title}\n\n`{cmd} --steps 10 --warmup 2 {Q}`\n\n")
        f.write("| kernel | HBM MB / launch | MB / frame | avg us (trace) | TB/s (counter bytes) | waves | VALU / wave | SALU / wave | LDS / wave | VMEM rd / wave | "
                "VMEM wr / wave | wait-any % of wave cycles | L2 hit % | L2 atomics |\n|" + "---|" * 14 + "\n")
        traffic[wl] = {}
        for k, d in sorted(rows.items()):
            g = lambda n: d.get(n, float("nan"))
            hbm = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
            us = avg_us.get(key, {}).get(k, float("nan"))
            w = g("SQ_WAVES")
            f.write(f"| `{k}` | {hbm/1e6:.2f} | {hbm/fpl/1e6:.3f} | {us:.2f} | {hbm/us/1e6:.2f} | {w:.0f} | {g('SQ_INSTS_VALU')/w:.0f} | {g('SQ_INSTS_SALU')/w:.0f} | "
                    f"{g('SQ_INSTS_LDS')/w:.0f} | {g('SQ_INSTS_VMEM_RD')/w:.1f} | {g('SQ_INSTS_VMEM_WR')/w:.1f} | {100*g('SQ_WAIT_ANY')/g('SQ_WAVE_CYCLES'):.0f} | "
                    f"{100*g('TCC_HIT_sum')/max(g('TCC_REQ_sum'),1):.0f} | {g('TCC_ATOMIC_sum'):.0f} |\n")
            lg = logical(k)
            if lg and hbm == hbm and (lg not in traffic[wl] or hbm > traffic[wl][lg]["hbm_bytes_per_launch"]):
                traffic[wl][lg] = {"kernel": k, "FETCH_SIZE_KB": round(g("FETCH_SIZE"), 1), "WRITE_SIZE_KB": round(g("WRITE_SIZE"), 1),
                                   "frames_per_launch": fpl, "hbm_bytes_per_launch": round(hbm), "hbm_bytes_per_frame": round(hbm / fpl)}
        f.write("\n| kernel | counter | avg per dispatch |\n|---|---|---|\n")
        for k, d in sorted(rows.items()):
            for cn, v in sorted(d.items()):
                f.write(f"| `{k}` | {cn} | {v:.1f} |\n")
        f.write("\n")
base = os.path.join(ROOT, "profiles", "pmc_traffic.json")
old = json.load(open(base)) if os.path.exists(base) else {}
old.update({k: v for k, v in traffic.items() if v})
old["_note_r0" + rnd] = ("projector_groups / camera_groups / projector_groups_esl / projector_sharded (merge = columns) / projector_sharded_keys / projector: round %s (tools/r0%s/make_profiles.sh, tag %s), "
                         "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, 2*FETCH_SIZE + WRITE_SIZE per the gfx950 calibration; summary profiles/%s_pmc.md" % (rnd, rnd, tag, tag))
json.dump(old, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(out)), {k: list(v) for k, v in traffic.items()})
