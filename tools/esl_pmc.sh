#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-esl_pmc}; mkdir -p $OUT
export ONLY=single
CMD="python tools/esl_groups.py"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT -o pmc_$i -- $CMD > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed: $set"
done
for f in $OUT/*.db; do python tools/rocprof_summary.py $f > ${f%.db}_summary.md 2>&1; done
grep -h -E "k_frame_proj|k_scatter|k_minmax|^\| kernel|^\|---" $OUT/*_summary.md | cut -c1-400
