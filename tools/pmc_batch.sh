#!/bin/bash
# PMC passes over the SATURATED multi-frame launches (60 x C-1M in one grid per kernel): what bounds K1 / K2 / K0 when the
# chip is full, as opposed to a single frame's latency chain.  Each --pmc group is its own run with --kernel-trace only;
# every run is bounded by its own timeout.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-pmc_batch}; mkdir -p $OUT
CMD="python tools/batch_probe.py 60 4 ${XM_PROBE_MODE:-0}"
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace failed"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT -o pmc_$i -- $CMD > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed: $set"
done
for f in $OUT/*.db; do python tools/rocprof_summary.py $f > ${f%.db}_summary.md 2>&1; done
ls $OUT
