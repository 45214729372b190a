#!/bin/bash
# memory-side counters of the group kernels (latencies, outstanding requests): each --pmc group its own run, --kernel-trace only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc_mem
python3 -c "import shutil, sys; shutil.rmtree(sys.argv[1], ignore_errors=True)" "$OUT"; mkdir -p "$OUT"
export XM_BENCH_PREWARM_S=0.05
Q="--no-cpu-baseline --no-other-modes --no-host-path --groups-in-flight 1 --steps 10 --warmup 2"
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_EA_RDREQ_LEVEL_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_RD_UNCACHED_32B_sum" \
           "TCC_EA_WRREQ_LEVEL_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA_ATOMIC_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_LOAD_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT -o mem_$i -- python bench.py $Q > $OUT/mem_$i.log 2>&1 || echo "pass $i failed: $set"
done
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys
from collections import defaultdict
rows = defaultdict(dict)
for db in sorted(glob.glob(os.path.join(sys.argv[1], "mem_*_results.db"))):
    try:
        for k, cn, v in sqlite3.connect(db).execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1, 2"):
            if "xm::k_" in k and ("batch" in k or "pipe" in k):
                rows[k.split("(")[0].replace("void ", "")][cn] = v
    except sqlite3.Error as e:
        print(db, e)
with open(os.path.join(sys.argv[1], "pmc_mem.md"), "w") as f:
    for k, d in sorted(rows.items()):
        f.write("## %s\n\n| counter | avg per dispatch |\n|---|---|\n" % k)
        for cn, v in sorted(d.items()):
            f.write("| %s | %.1f |\n" % (cn, v))
        f.write("\n")
print(open(os.path.join(sys.argv[1], "pmc_mem.md")).read()[:6000])
PY
python3 -c "import glob, os, sys; [os.remove(f) for p in ('*.db', '*.csv') for f in glob.glob(os.path.join(sys.argv[1], p))]" "$OUT"
