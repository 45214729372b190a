#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
Q="--esl --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs"
for G in 4 6 8; do for K in 8 3 2 1; do
  python bench.py $Q --groups-in-flight $G --lib-option XM_K2_PER_CU=$K 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('groups in flight $G K2 per CU $K :', d.get('value'), 'Mev/s', d.get('ms_per_step'), 'ms/step', d.get('error'))"
done; done
