#!/usr/bin/env python3
"""Timeline (kernels + memory copies, by queue / stream) of a window of a rocprofv3 results db.
    python tools/r04/ingest_timeline.py <results.db> [from_fraction=0.8 | -microseconds_before_the_end] [window_us=1500]
Prints the schema of the copy table the first time, so that the columns are known."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
win = float(sys.argv[3]) if len(sys.argv) > 3 else 1500.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]

rows = []
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]

qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
for r in db.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels"):
    rows.append((r[1], r[2], "K", r[0].split("(")[0].replace("xm::", "")[:44], r[3] if qcol else -1))
for t in ("memory_copies", "memory_copy"):
    if t in tabs:
        mc = [r[1] for r in db.execute(f"pragma table_info({t})")]

        name = "name" if "name" in mc else mc[0]
        size = "size" if "size" in mc else None
        for r in db.execute(f"select {name}, start, end{', ' + size if size else ''} from {t}"):
            rows.append((r[1], r[2], "C", f"{r[0]} {r[3] if size else ''}", -2))
        break
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
a = t0 + (t1 - t0) * frac if frac >= 0 else t1 + frac * 1e3  # (negative: microseconds before the last record)
print(f"span {(t1 - t0) / 1e6:.1f} ms, {len(rows)} records; window from {frac:.2f} of it, {win:.0f} us")
for s, e, kind, nm, q in rows:
    if a <= s <= a + win * 1e3:
        print(f"{(s - a) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {kind} q{q:<3} {nm}")
