#!/bin/bash
# A/B/C: default build vs build/<libs>: ESL single frames, K2 60-frame launch, bench (alternating)
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  echo "=== ${v:-default}"
  if [ -n "$v" ]; then export XM_LIB=build/$v; else unset XM_LIB; fi
  ONLY=single timeout 200 python tools/esl_groups.py 2>&1 | grep -E "single|rror"
  bash tools/batch_trace.sh bt_${v:-def} 2>&1 | grep -E "k_frame" | cut -c1-120
done
for i in 1 2 3; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export XM_LIB=build/$v; else unset XM_LIB; fi
    timeout 100 python bench.py --no-cpu-baseline --no-other-modes --no-host-path 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-default}', d['value'], d['ms_per_step'])"
  done
done
