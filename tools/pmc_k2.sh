#!/bin/bash
# PMC passes over the saturated multi-frame launches (library defaults: column tiles), K2 + K1 rows only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-pmc_k2}; mkdir -p $OUT
CMD="python tools/batch_probe.py 60 4 2"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT -o pmc_$i -- $CMD > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed: $set"
done
for f in $OUT/*.db; do python tools/rocprof_summary.py $f > ${f%.db}_summary.md 2>&1; done
grep -h -E "k_frame_proj|k_scatter_cols|k_cols_bounds" $OUT/*_summary.md | cut -c1-200
