#!/usr/bin/env python3
"""Turn the rocprofv3 databases of tools/make_profiles.sh (merged back under gpurun_out/<tag>/) into the tracked
summaries under profiles/: <tag>_kernel_trace.md, <tag>_pmc.md and pmc_traffic.json (read by bench.py)."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
# on the GPU box the databases are too large to travel back (64 MiB cap): `--to-scratch` writes the summaries next to them,
# under gpurun_out/<tag>_profiles/, from where they are copied into profiles/ once merged back
out = os.path.join(ROOT, "gpurun_out", tag + "_profiles") if "--to-scratch" in sys.argv else os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "")


def logical(name):
    if "k_minmax" in name:
        return "k_minmax"
    if "k_scatter" in name:
        return "k_scatter"
    if "k_frame" in name:
        return "k_frame"
    return None


def trace_table(db):
    c = sqlite3.connect(db)
    agg = defaultdict(list)
    for name, d in c.execute("select name, end - start from kernels"):
        agg[name].append(d)
    total = sum(sum(v) for v in agg.values())
    lines = ["| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"]
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| `{short(name)}` | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | "
                     f"{sum(v) / 1e6:.3f} | {100 * sum(v) / total:.1f} |")
    return "\n".join(lines)


def pmc_rows(pattern):
    rows = {}
    for db in sorted(glob.glob(os.path.join(src, pattern))):
        c = sqlite3.connect(db)
        try:
            for k, cn, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1, 2"):
                if "xm::" in k and "reset" not in k and "build_dlut" not in k:
                    rows[(short(k), cn)] = v
        except sqlite3.Error:
            pass
    return rows


with open(os.path.join(out, f"{tag}_kernel_trace.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of bench.py (MI355X)\n\n")
    for key, title, cmd in (("trace_proj", "projector view, 1 slot, library defaults (kernels back to back, no overlap; no K0: verified (t[0], t[n-1]) shortcut)", "python bench.py --slots 1 --steps 200 --warmup 20 --no-cpu-baseline --no-other-modes --no-host-path"),
                            ("trace_general", "projector view, 1 slot, XM_FLAG_GENERAL (K0 on every frame)", "... --general"),
                            ("trace_cam", "camera view, 1 slot", "... --camera-perspective"),
                            ("trace_pipe8", "projector view, 4 frames in flight (round-1 name)", "python bench.py --steps 400 --no-cpu-baseline"),
                            ("trace_pipe", "projector view, 4 frames in flight (the default bench configuration; under the profiler the launches no longer overlap)", "python bench.py --steps 400 --no-cpu-baseline --no-other-modes --no-host-path"),
                            ("trace_batch60", "multi-frame launches: 60 x C-1M frames per kernel launch (grid = frames x tiles), general path", "python tools/batch_probe.py 60 4"),
                            ("trace_batch60_sorted", "multi-frame launches, 60 frames, declared time-sorted (no K0, 64-bit key frame)", "python tools/batch_probe.py 60 4 1"),
                            ("trace_batch60_default", "multi-frame launches, 60 frames, library defaults (verified shortcut + compact 32-bit key frame)", "python tools/batch_probe.py 60 4 2")):
        db = os.path.join(src, f"{key}_results.db")
        if os.path.exists(db):
            f.write(f"## {title}\n\n`rocprofv3 --kernel-trace --stats -- {cmd}`\n\n{trace_table(db)}\n\n")
    for j in ("bench_default", "bench_steps20", "bench_steps8000", "bench_camera", "bench_graph60", "bench_graph60_slots8", "bench_sharded"):
        p = os.path.join(src, j + ".json")
        if os.path.exists(p) and os.path.getsize(p):
            f.write(f"## {j}.json (un-profiled run on the same box)\n\n```json\n{open(p).read().strip()}\n```\n\n")
    sp = os.path.join(src, "scale_probe.txt")
    if os.path.exists(sp):
        f.write("## tools/scale_probe.py (HIP-event time per kernel vs events per frame, 1 slot)\n\n```\n" + open(sp).read() + "```\n")

traffic = {}
with open(os.path.join(out, f"{tag}_pmc.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 PMC counters per kernel (averages per dispatch; one --pmc group per run, `tools/make_profiles.sh`)\n\n")
    f.write("FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read\n"
            "(MI355X_MICROARCH.md section HBM; confirmed here: k_minmax reads exactly 8.0 MB of t and shows ~3.9 MB), so\n"
            "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE.\n\n")
    for view, pat in (("projector", "pmc_proj_*_results.db"), ("projector_general", "pmc_gen_*_results.db"),
                      ("camera", "pmc_cam_*_results.db"), ("batch60", "pmc_batch_*_results.db"),
                      ("batch60_default", "pmc_bdef_*_results.db")):
        rows = pmc_rows(pat)
        if not rows:
            continue
        f.write(f"## {view} view\n\n| kernel | counter | avg per dispatch |\n|---|---|---|\n")
        for (k, cn), v in sorted(rows.items()):
            f.write(f"| `{k}` | {cn} | {v:.1f} |\n")
        f.write("\n")
        traffic[view] = {}
        kernels = sorted({k for (k, _) in rows})
        for k in kernels:
            lg = logical(k)
            fs, ws = rows.get((k, "FETCH_SIZE")), rows.get((k, "WRITE_SIZE"))
            if lg and fs is not None and ws is not None:
                traffic[view][lg] = {"kernel": k, "FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1),
                                     "hbm_bytes_per_launch": round((2 * fs + ws) * 1024)}
if "projector" in traffic and "projector_general" in traffic:  # K0 only runs on the general path: its traffic comes from there
    for k, v in traffic["projector_general"].items():
        traffic["projector"].setdefault(k, v)
if traffic:
    traffic["_note"] = ("per-launch HBM bytes of the C-1M bench workload from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                        "2*FETCH_SIZE + WRITE_SIZE per the gfx950 calibration; source gpurun_out/%s, summary profiles/%s_pmc.md" % (tag, tag))
    with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
print("wrote", os.listdir(out))
