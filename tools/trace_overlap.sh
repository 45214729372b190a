#!/bin/bash
# Experiment: kernel start/end timestamps of the pipelined bench (rocprofv3 --kernel-trace) -> durations under overlap,
# concurrency histogram, gaps between dependent kernels of one frame.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/overlap; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o t -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline $BENCH_ARGS > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-160
python - <<'PY'
import sqlite3, glob, collections
db = glob.glob('gpurun_out/overlap/t_results.db')[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall() if 'stream_id' in cols else \
       [(r[0], r[1], r[2], r[3], 0) for r in c.execute("select name, start, end, queue_id from kernels order by start")]
rows = [r for r in rows if 'xm::' in r[0] and ('minmax' in r[0] or 'scatter' in r[0] or 'frame' in r[0])]
rows = rows[len(rows) // 3:]           # steady state
def short(n): return 'K0' if 'minmax' in n else ('K1' if 'scatter' in n else 'K2')
dur = collections.defaultdict(list)
for n, s, e, q, st in rows: dur[short(n)].append((e - s) / 1e3)
for k in sorted(dur): 
    v = sorted(dur[k]); print(f"{k}: n={len(v)} mean {sum(v)/len(v):6.2f} us  p10 {v[len(v)//10]:6.2f}  p50 {v[len(v)//2]:6.2f}  p90 {v[9*len(v)//10]:6.2f}")
t0, t1 = rows[0][1], max(r[2] for r in rows)
ev = []
for n, s, e, q, st in rows: ev += [(s, 1, short(n)), (e, -1, short(n))]
ev.sort()
hist = collections.Counter(); cur = 0; last = ev[0][0]
kcur = collections.Counter(); khist = collections.Counter()
for t, d, k in ev:
    hist[cur] += t - last
    khist[tuple(sorted(kcur.elements()))] += t - last
    last = t; cur += d; kcur[k] += d
tot = sum(hist.values())
print("concurrency (kernels running at once): " + "  ".join(f"{k}:{100*v/tot:.0f}%" for k, v in sorted(hist.items())))
print("top mixes: " + "  ".join(f"{''.join(k) or '-'}:{100*v/tot:.0f}%" for k, v in khist.most_common(8)))
print(f"frames {len(dur['K1'])} in {(t1-t0)/1e3:.0f} us -> {(t1-t0)/1e3/len(dur['K1']):.2f} us/frame; queues used: {len(set(r[3] for r in rows))}")
# gaps inside a queue
byq = collections.defaultdict(list)
for n, s, e, q, st in rows: byq[q].append((s, e, short(n)))
gaps = collections.defaultdict(list)
for q, lst in byq.items():
    lst.sort()
    for (s0, e0, k0), (s1, e1, k1) in zip(lst, lst[1:]): gaps[k0 + '->' + k1].append((s1 - e0) / 1e3)
for k, v in sorted(gaps.items()): print(f"gap {k}: mean {sum(v)/len(v):6.2f} us (n={len(v)})")
PY
