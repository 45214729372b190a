#!/bin/bash
# round 3: kernel trace + PMC passes of one command (each --pmc group its own run, --kernel-trace only, bounded by a timeout);
# summary -> gpurun_out/<tag>/summary.md (+ traffic.json: HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, KB units on gfx950)
#   bash tools/r03/pmc.sh <tag> <frames per launch> <command ...>
TAG=$1; FPL=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- "$@" > $OUT/trace.log 2>&1 || echo "trace failed"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum"; do
  i=$((i+1))
  [ -n "$PMC_ONLY" ] && [ $i -gt $PMC_ONLY ] && break
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT -o pmc_$i -- "$@" > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed: $set"
done
python tools/r03/pmc_summary.py $OUT $FPL > $OUT/summary.md 2>&1
rm -f $OUT/*.db $OUT/*.csv
cat $OUT/summary.md | cut -c1-230
