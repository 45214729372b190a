#!/bin/bash
# A/B of build/<lib> against the default build, alternating runs in one session: bash tools/ab.sh <lib.so> [bench flags]
cd $GRAFT_REPO_ROOT
LIBV=build/$1; shift
for i in 1 2 3; do
  for which in var def; do
    if [ $which = var ]; then export XM_LIB=$LIBV; else unset XM_LIB; fi
    timeout 100 python bench.py --no-cpu-baseline --no-other-modes --no-host-path "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'] if 'kernel_us' in d['roofline'] else '')"
  done
done
