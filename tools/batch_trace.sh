#!/bin/bash
# Kernel trace of the SATURATED multi-frame launches (60 x C-1M per grid), library defaults, for the library in XM_LIB (or the
# default build) -- per-frame cost of K1 / K2 when the chip is full:  bash tools/batch_trace.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-bt}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
env "$@" timeout 150 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python tools/batch_probe.py 60 6 2 > $OUT/trace.log 2>&1 || echo "trace failed"
for f in $OUT/*.db; do python tools/rocprof_summary.py $f > ${f%.db}_summary.md 2>&1; done
grep -E "k_scatter|k_frame|k_minmax|k_cols" $OUT/*_summary.md | cut -c1-220
