#!/bin/bash
# One K2 at a time (XM_K2_CHAIN=1) at K blocks per CU, the following groups' K0b / K1 in what it leaves.  ESL-like and C-1M groups.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs"
for mode in ${MODES:---esl c1m}; do m=$mode; [ $mode = c1m ] && m=""; for C in 0 1; do for K in ${@:-8 5 4 3}; do
  [ $C = 0 ] && [ $K != 8 ] && continue
  python bench.py $m $Q --lib-option XM_K2_CHAIN=$C --lib-option XM_K2_PER_CU=$K $XM_SWEEP_OPTS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$mode chain $C K2 per CU $K :', d.get('value'), 'Mev/s', d.get('ms_per_step'), 'ms/step', d.get('parity',{}).get('group_last_frame_depth_bit_exact', d.get('parity',{}).get('depth_bit_exact')), d.get('error'))"
done; done; done
