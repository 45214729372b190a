#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite .db or *_kernel_trace.csv) into a small markdown table
(per-kernel count / avg / min / max / total), for committing under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_trace.md
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, end - start from kernels").fetchall()
    pmc = defaultdict(lambda: defaultdict(list))
    try:
        for name, counter, value in c.execute(
                "select k.name, p.counter_name, p.value from counters_collection p join kernels k on p.dispatch_id = k.dispatch_id"):
            pmc[name][counter].append(value)
    except sqlite3.Error:
        pass
    return rows, pmc


def from_csv(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return rows, {}


def main():
    path = sys.argv[1]
    rows, pmc = from_db(path) if path.endswith(".db") else from_csv(path)
    agg = defaultdict(list)
    for name, d in rows:
        agg[name].append(d)
    total = sum(sum(v) for v in agg.values())
    print(f"source: `{path}`\n")
    print("| kernel | calls | avg us | min us | max us | total ms | % |")
    print("|---|---|---|---|---|---|---|")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name.split("(")[0].replace("void ", "")
        print(f"| `{short}` | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | "
              f"{sum(v) / 1e6:.3f} | {100 * sum(v) / total:.1f} |")
    if pmc:
        print("\n| kernel | counter | avg per dispatch |")
        print("|---|---|---|")
        for name, cs in pmc.items():
            short = name.split("(")[0].replace("void ", "")
            for cname, vals in sorted(cs.items()):
                print(f"| `{short}` | {cname} | {sum(vals) / len(vals):.1f} |")


if __name__ == "__main__":
    main()
