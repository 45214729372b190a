#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
Q="--esl --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs"
for rep in 1 2 3; do for K in 8 5 4 2; do
  python bench.py $Q --lib-option XM_K2_PER_CU=$K 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('rep $rep K2 per CU $K :', d.get('value'), 'Mev/s', d.get('ms_per_step'), 'ms/step', d.get('error'))"
done; done
