#!/bin/bash
# Kernel trace of ESL-like frames (tools/esl_groups.py): bash tools/esl_trace.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-esl}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
env "$@" timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python tools/esl_groups.py > $OUT/trace.log 2>&1 || echo "trace failed"
for f in $OUT/*.db; do python tools/rocprof_summary.py $f > ${f%.db}_summary.md 2>&1; done
grep -E "k_scatter|k_frame|k_minmax|k_cols|k_reset|k_clear" $OUT/*_summary.md | cut -c1-200
tail -4 $OUT/trace.log
