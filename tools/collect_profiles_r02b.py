#!/usr/bin/env python3
"""Turn the rocprofv3 databases of tools/make_profiles_r02b.sh into the tracked summaries under profiles/:
<tag>_kernel_trace.md, <tag>_pmc.md, and the `*_groups` entries of pmc_traffic.json (read by bench.py).  On the GPU box
(`--to-scratch`) the summaries go to gpurun_out/<tag>_profiles/, from where they are copied into profiles/ once merged back."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02b"
src = os.path.join(ROOT, "gpurun_out", tag)
out = os.path.join(ROOT, "gpurun_out", tag + "_profiles") if "--to-scratch" in sys.argv else os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
FRAMES_PER_LAUNCH = 32


def short(name):
    return name.split("(")[0].replace("void ", "")


def logical(name):
    if "k_minmax" in name or "k_cols_bounds" in name:
        return "k_minmax"  # the helper pass in front of K1
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


Q = "--no-cpu-baseline --no-other-modes --no-host-path"
with open(os.path.join(out, f"{tag}_kernel_trace.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of bench.py (MI355X), end of round 2\n\n"
            "A bench step = one group of 32 C-1M frames through xm_process_batch: ONE launch each of k_cols_bounds_batch (K0b),\n"
            "k_scatter_cols_batch (K1, column tiles) and k_frame_proj_tiled_batch<2> (K2 on the u16 frame), grid = frames x tiles.\n"
            "Per-frame cost = avg us / 32.\n\n")
    for key, title, cmd in (
            ("trace_groups", "the default bench command (projector view, 3 groups in flight)", f"python bench.py {Q}"),
            ("trace_serial", "one group at a time (launches back to back, nothing overlaps): what bench.py's roofline pass times", f"python bench.py --groups-in-flight 1 {Q}"),
            ("trace_single", "one frame per call, 1 slot (single-frame launches: compact 32-bit key frame path)", f"python bench.py --batch 0 --slots 1 --steps 200 --warmup 20 {Q}"),
            ("trace_cam", "camera view, one group at a time", f"python bench.py --groups-in-flight 1 --camera-perspective {Q}"),
            ("trace_batch60", "60 frames per launch (chip saturated for ~0.25 ms per kernel)", "python tools/batch_probe.py 60 4 2")):
        db = os.path.join(src, f"{key}_results.db")
        if os.path.exists(db):
            f.write(f"## {title}\n\n`rocprofv3 --kernel-trace --stats -- {cmd}`\n\n{trace_table(db)}\n\n")
    for j in ("bench_default", "bench_steps20", "bench_steps2000", "bench_one_frame_per_call", "bench_camera", "bench_graph60",
              "bench_sharded", "bench_esl"):
        p = os.path.join(src, j + ".json")
        if os.path.exists(p) and os.path.getsize(p):
            f.write(f"## {j}.json (un-profiled run on the same box)\n\n```json\n{open(p).read().strip().splitlines()[-1]}\n```\n\n")

traffic = {}
with open(os.path.join(out, f"{tag}_pmc.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 PMC counters per kernel (averages per dispatch of a 32-frame launch; one --pmc group per run)\n\n"
            "FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read\n"
            "(MI355X_MICROARCH.md section HBM), so HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE.\n\n")
    for view, pat in (("projector_groups", "pmc_groups_*_results.db"), ("camera_groups", "pmc_camg_*_results.db")):
        rows = pmc_rows(pat)
        if not rows:
            continue
        f.write(f"## {view}\n\n| kernel | counter | avg per dispatch |\n|---|---|---|\n")
        for (k, cn), v in sorted(rows.items()):
            f.write(f"| `{k}` | {cn} | {v:.1f} |\n")
        f.write("\n")
        traffic[view] = {}
        for k in sorted({k for (k, _) in rows}):
            lg = logical(k)
            fs, ws = rows.get((k, "FETCH_SIZE")), rows.get((k, "WRITE_SIZE"))
            if lg and "batch" in k and fs is not None and ws is not None:
                b = round((2 * fs + ws) * 1024)
                traffic[view][lg] = {"kernel": k, "FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1), "frames_per_launch": FRAMES_PER_LAUNCH,
                                     "hbm_bytes_per_launch": b, "hbm_bytes_per_frame": round(b / FRAMES_PER_LAUNCH)}
if traffic:
    for base in (os.path.join(ROOT, "profiles", "pmc_traffic.json"),):
        old = {}
        if os.path.exists(base):
            old = json.load(open(base))
        old.update(traffic)
        old["_note_groups"] = ("*_groups: 32-frame launches (bench.py default since the end of round 2), rocprofv3 --pmc FETCH_SIZE / "
                               "WRITE_SIZE in separate passes, 2*FETCH_SIZE + WRITE_SIZE per the gfx950 calibration; source gpurun_out/%s, "
                               "summary profiles/%s_pmc.md" % (tag, tag))
        with open(os.path.join(out, "pmc_traffic.json"), "w") as g:
            json.dump(old, g, indent=1)
print("wrote", os.listdir(out))
