#!/bin/bash
# A/B of K2 pixels-per-thread builds: ESL-like single frames, C-1M group traces, bench line
cd $GRAFT_REPO_ROOT
export ONLY=single
for v in "" "$@"; do
  echo "=== ${v:-default}"
  [ -n "$v" ] && export XM_LIB=build/$v
  timeout 200 python tools/esl_groups.py 2>&1 | grep -E "single|rror"
  bash tools/batch_trace.sh bt_${v:-def} 2>&1 | grep -E "k_frame|k_scatter_cols" | cut -c1-160
  timeout 200 python bench.py --no-cpu-baseline --no-other-modes --no-host-path 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('parity', {}).get('depth_max_rel_err'))"
done
