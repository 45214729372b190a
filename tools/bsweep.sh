#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "32 3" "64 2" "48 3" "32 4" "16 4"; do
  set -- $cfg
  timeout 150 python bench.py --no-cpu-baseline --no-other-modes --no-host-path --batch $1 --groups-in-flight $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$1 G=$2', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
