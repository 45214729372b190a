#!/bin/bash
# Experiment: L2<->fabric traffic per launch of the hot kernels (FETCH_SIZE / WRITE_SIZE in separate passes; gfx950: HBM bytes
# = 2 x FETCH_SIZE + WRITE_SIZE, KB units).  Env passes through (e.g. XM_K2_FLAGS=1); BENCH_ARGS adds bench flags.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/traffic; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT -o p$i -- python bench.py --slots 1 --steps 40 --warmup 5 --no-cpu-baseline $BENCH_ARGS > /dev/null 2>&1
done
python - <<'PY'
import sqlite3, glob
rows = {}
for db in sorted(glob.glob('gpurun_out/traffic/p*_results.db')):
    c = sqlite3.connect(db)
    for k, cn, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1,2"):
        if 'xm::' in k and 'reset' not in k and 'dlut' not in k and 'k2_tables' not in k:
            rows.setdefault(k.split('(')[0].replace('void xm::', '')[:30], {})[cn] = v
tot = 0
for k, d in rows.items():
    hbm = (2 * d.get('FETCH_SIZE', 0) + d.get('WRITE_SIZE', 0)) / 1024
    tot += hbm
    print(f"{k:32s} fetch {d.get('FETCH_SIZE',0)/1024:7.2f} MB(x2)  write {d.get('WRITE_SIZE',0)/1024:6.2f} MB  -> {hbm:6.2f} MB/launch   L2 req {d.get('TCC_REQ_sum',0):9.0f} hit {d.get('TCC_HIT_sum',0):9.0f} miss {d.get('TCC_MISS_sum',0):9.0f} atomic {d.get('TCC_ATOMIC_sum',0):9.0f}")
print(f"total {tot:.1f} MB per frame")
PY
