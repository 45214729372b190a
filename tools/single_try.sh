#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for c in 1 2; do
  XM_COLS=$c timeout 150 python bench.py --no-cpu-baseline --no-other-modes --no-host-path --batch 0 --steps 2000 --warmup 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('XM_COLS=$c one frame per call', d['value'], d['ms_per_step'], d['config'].get('k1_paths_frames'))"
done
done
