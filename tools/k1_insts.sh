#!/bin/bash
# dynamic instruction counts per wave of the hot kernels (one PMC pass) + isolated/pipelined timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/insts; mkdir -p gpurun_out/insts
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d gpurun_out/insts -o p -- python bench.py --slots 1 --steps 40 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import sqlite3
c=sqlite3.connect('gpurun_out/insts/p_results.db')
rows={}
for k,cn,v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1,2"):
    if 'xm::' in k and 'reset' not in k and 'dlut' not in k: rows.setdefault(k.split('(')[0].replace('void xm::','')[:32],{})[cn]=v
for k,d in rows.items():
    w=d.get('SQ_WAVES',1)
    print(k, 'waves',int(w), ' per wave: VALU %.0f SALU %.0f LDS %.0f VMEM %.0f SMEM %.0f'%(d.get('SQ_INSTS_VALU',0)/w,d.get('SQ_INSTS_SALU',0)/w,d.get('SQ_INSTS_LDS',0)/w,(d.get('SQ_INSTS_VMEM_RD',0)+d.get('SQ_INSTS_VMEM_WR',0))/w,d.get('SQ_INSTS_SMEM',0)/w))
PY
for s in 1 8; do python bench.py --no-cpu-baseline --slots $s 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slots',d['config']['frames_in_flight'],d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['parity']['depth_bit_exact'],d['parity']['bgr_equal'])"; done
