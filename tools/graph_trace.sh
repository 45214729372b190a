#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/graph_trace; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python bench.py --graph --no-cpu-baseline > $OUT/log 2>&1
python tools/rocprof_summary.py $OUT/t_results.db 2>/dev/null | grep -E "^\| " | head -16 | cut -c1-170
