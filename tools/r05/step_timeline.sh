#!/bin/bash
# Where does the pipelined step go?  rocprofv3 --kernel-trace of the default bench (four groups in flight), then: over the timed
# region's K0b / K1 / K2 dispatches the union of busy time, the overlap between kernels and the idle gaps between them.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05/steptl; rm -rf $OUT; mkdir -p $OUT
XM_BENCH_PREWARM_S=0.05 timeout 300 rocprofv3 --kernel-trace -d $OUT -o tl -- python bench.py ${@:-} --steps 40 --warmup 5 --single-block --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs > $OUT/log.txt 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/**/tl_results.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
print("tables:", kd[:4])
rows = con.execute("select name, start, end from kernels order by start").fetchall() if "kernels" in tabs else []
rows = [(n, s, e) for n, s, e in rows if ("k_cols_bounds_batch" in n or "k_scatter_cols_batch" in n or "k_scatter_own_batch" in n or "k_frame_proj_pipe" in n)]
rows = rows[len(rows) // 2:]  # the second half: the timed region, steady state
t0, t1 = rows[0][1], max(e for _, _, e in rows)
busy, cur_s, cur_e, overlap = 0, rows[0][1], rows[0][2], 0
gaps = []
for n, s, e in rows[1:]:
    if s <= cur_e:
        overlap += min(e, cur_e) - s
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
busy += cur_e - cur_s
kinds = {}
for n, s, e in rows:
    k = "K0b" if "bounds" in n else "K1" if "scatter" in n else "K2"
    kinds.setdefault(k, []).append(e - s)
groups = len(kinds.get("K1", []))
print("groups %d  span %.1f us per group  busy %.1f  idle %.1f (%d gaps, mean %.2f us)  overlapped kernel time %.1f us per group" % (
    groups, (t1 - t0) / 1e3 / groups, busy / 1e3 / groups, sum(gaps) / 1e3 / groups, len(gaps), (sum(gaps) / max(len(gaps), 1)) / 1e3, overlap / 1e3 / groups))
for k, v in kinds.items():
    print("  %s mean %.2f us (n %d)" % (k, sum(v) / len(v) / 1e3, len(v)))
# what runs when: time per group by the set of kernel kinds in flight
ev = []
for n, s, e in rows:
    k = "K0b" if "bounds" in n else "K1" if "scatter" in n else "K2"
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
cnt = {"K0b": 0, "K1": 0, "K2": 0}
acc, last = {}, ev[0][0]
for t, d, k in ev:
    key = "+".join("%s x%d" % (q, cnt[q]) for q in ("K0b", "K1", "K2") if cnt[q]) or "idle"
    acc[key] = acc.get(key, 0) + (t - last)
    last = t
    cnt[k] += d
for key, v in sorted(acc.items(), key=lambda kv: -kv[1])[:12]:
    print("  %-28s %7.2f us per group" % (key, v / 1e3 / groups))
PY
rm -rf $OUT
