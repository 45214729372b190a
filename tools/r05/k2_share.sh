#!/bin/bash
# Does the next group's K1 run beside this group's K2 when the (persistent) K2 leaves room on the CUs?  XM_K2_PER_CU = blocks of
# the pipelined K2 per CU (default: as many as the LDS holds, 3 on both rigs).  ESL-like groups and the headline C-1M groups,
# pipelined step (4 groups in flight) + one group at a time.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs"
for mode in "--esl" ""; do for K in ${@:-8 2 1}; do for P in 2 1; do
  [ -z "$mode" ] && [ $P = 1 ] && continue
  python bench.py $mode $Q --lib-option XM_K2_PER_CU=$K --lib-option XM_OWN_ROW_PASSES=$P 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d.get('roofline') or {}
print('${mode:-c1m} K2 per CU $K own passes $P :', d.get('value'), 'Mev/s', d.get('ms_per_step'), 'ms/step  K0b/K1/K2 us', r.get('avg_launch_us'), d.get('error'))"
done; done; done
