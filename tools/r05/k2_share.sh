#!/bin/bash
# Does the next group's K1 run beside this group's K2 when the (persistent) K2 leaves room on the CUs?  XM_K2_PER_CU = blocks of
# the pipelined K2 per CU at most (default 8: as many as LDS and registers hold -- 6 on both rigs).  ESL-like and headline C-1M
# groups: the pipelined step (4 groups in flight) and one group at a time, REPS times each (the boxes have two clock states).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs"
for rep in $(seq 1 ${REPS:-2}); do for mode in ${MODES:---esl c1m}; do m=$mode; [ $mode = c1m ] && m=""; for K in ${@:-8 5 4 2}; do for gif in 0 1; do
  extra=""; [ $gif = 1 ] && extra="--groups-in-flight 1 --steps 10 --warmup 2"
  python bench.py $m $Q $extra --lib-option XM_K2_PER_CU=$K 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d.get('roofline') or {}
print('rep $rep $mode K2 per CU $K one-at-a-time $gif :', d.get('value'), 'Mev/s', d.get('ms_per_step'), 'ms/step  K0b/K1/K2 us', r.get('avg_launch_us'), d.get('error'))"
done; done; done; done
