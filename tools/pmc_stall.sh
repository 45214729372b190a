#!/bin/bash
# Experiment: where do the waves of the hot kernels spend their cycles? (isolated kernels, slots=1; one PMC pass per set)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/stall; mkdir -p gpurun_out/stall
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/stall -o p$i -- python bench.py --slots 1 --steps 30 --warmup 5 --no-cpu-baseline $BENCH_ARGS > /dev/null 2>&1
done
python - <<'PY'
import sqlite3, glob
rows = {}
for db in sorted(glob.glob('gpurun_out/stall/p*_results.db')):
    c = sqlite3.connect(db)
    for k, cn, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1,2"):
        if 'xm::' in k and 'reset' not in k and 'dlut' not in k and 'k2_tables' not in k:
            rows.setdefault(k.split('(')[0].replace('void xm::', '')[:28], {})[cn] = v
for k, d in rows.items():
    print(k)
    w = d.get('SQ_WAVES', 1)
    for cn in sorted(d):
        print(f"   {cn:28s} {d[cn]:14.0f}   per wave {d[cn]/w:10.1f}")
PY
