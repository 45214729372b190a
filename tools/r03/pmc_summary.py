#!/usr/bin/env python3
"""gpurun_out/<tag>/{trace,pmc_*}_results.db -> markdown: per-kernel durations + counters per dispatch (averages), HBM bytes per
launch and per frame (2 x FETCH_SIZE + WRITE_SIZE KB: the gfx950 correction of MI355X_MICROARCH.md), derived rates."""
import glob, json, os, sqlite3, sys
from collections import defaultdict
src, fpl = sys.argv[1], float(sys.argv[2])
def short(n): return n.split("(")[0].replace("void ", "")
skip = ("reset", "build_dlut", "k2_tables", "cols_check", "build_x_map", "rocclr")
dur = defaultdict(list)
for db in glob.glob(os.path.join(src, "trace*_results.db")):
    for name, d in sqlite3.connect(db).execute("select name, end - start from kernels"):
        dur[short(name)].append(d)
print("| kernel | calls | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if any(s in k for s in skip): continue
    print(f"| `{k}` | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | {sum(v)/1e6:.3f} |")
rows = defaultdict(dict)
pdur = defaultdict(list)
for db in sorted(glob.glob(os.path.join(src, "pmc_*_results.db"))):
    c = sqlite3.connect(db)
    try:
        for k, cn, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1, 2"):
            if "xm::" in k and not any(s in k for s in skip): rows[short(k)][cn] = v
    except sqlite3.Error as e:
        print("(", db, e, ")")
print("\n| kernel | counter | avg per dispatch |\n|---|---|---|")
traffic = {}
for k, d in rows.items():
    for cn, v in sorted(d.items()): print(f"| `{k}` | {cn} | {v:.1f} |")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        hbm = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        traffic[k] = {"hbm_bytes_per_launch": hbm, "hbm_bytes_per_frame": hbm / fpl, "fetch_KB": d["FETCH_SIZE"], "write_KB": d["WRITE_SIZE"]}
print("\n| kernel | HBM MB / launch | MB / frame | avg us (trace) | TB/s |\n|---|---|---|---|---|")
for k, t in traffic.items():
    us = sum(dur[k]) / len(dur[k]) / 1e3 if dur.get(k) else float("nan")
    print(f"| `{k}` | {t['hbm_bytes_per_launch']/1e6:.2f} | {t['hbm_bytes_per_frame']/1e6:.3f} | {us:.2f} | {t['hbm_bytes_per_launch']/us/1e6:.2f} |")
print("\n| kernel | waves | VALU / wave | SALU / wave | LDS / wave | VMEM rd / wave | VMEM wr / wave | VALU busy % | LDS busy % | wait-any % | LDS conflict % of LDS active | TA busy % |\n|---|---|---|---|---|---|---|---|---|---|---|---|")
for k, d in rows.items():
    w = d.get("SQ_WAVES")
    if not w: continue
    g = lambda n: d.get(n, float("nan"))
    busy = g("SQ_BUSY_CYCLES"); wc = g("SQ_WAVE_CYCLES")
    print(f"| `{k}` | {w:.0f} | {g('SQ_INSTS_VALU')/w:.0f} | {g('SQ_INSTS_SALU')/w:.0f} | {g('SQ_INSTS_LDS')/w:.0f} | {g('SQ_INSTS_VMEM_RD')/w:.1f} | {g('SQ_INSTS_VMEM_WR')/w:.1f} | "
          f"{100*g('SQ_ACTIVE_INST_VALU')/busy:.0f} | {100*g('SQ_ACTIVE_INST_LDS')/busy:.0f} | {100*g('SQ_WAIT_ANY')/wc:.0f} | {100*g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):.0f} | "
          f"{100*g('TA_TA_BUSY_sum')/max(g('GRBM_GUI_ACTIVE'),1)/256:.0f} |")
json.dump(traffic, open(os.path.join(src, "traffic.json"), "w"), indent=1)
