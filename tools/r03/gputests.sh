#!/bin/bash
# the -m gpu suite (or the files given), full log -> gpurun_out/r03_gputest.log, summary lines on stdout
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2700 python -m pytest ${@:-tests} -x -q -m gpu > gpurun_out/r03_gputest.log 2>&1
grep -E "passed|failed|error|Error|assert" gpurun_out/r03_gputest.log | tail -15
