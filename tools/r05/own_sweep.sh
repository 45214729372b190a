#!/bin/bash
# Owner tiles on the ESL-like rig: tile width sweep (XM_OWN_W), per-kernel times of one group at a time + the pipelined step.
#   bash tools/r05/own_sweep.sh [W ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
Q="--esl --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs"
for W in ${@:-8 4 12 16 24 32}; do
  for gif in 1 0; do
    extra=""; [ $gif = 1 ] && extra="--groups-in-flight 1 --steps 10 --warmup 2"
    python bench.py $Q $extra --lib-option XM_OWN_W=$W $XM_SWEEP_OPTS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
if 'error' in d: print('W $W', d); sys.exit()
r=d['roofline']; g=d['config']['k1_geometry']
print('W $W gif $gif :', d['value'], 'Mev/s', d['ms_per_step'], 'ms/step  K0b/K1/K2 us', r.get('avg_launch_us'), 'geom', g, 'parity', d['parity'].get('group_first_frame_depth_bit_exact'), d['parity'].get('group_last_frame_depth_bit_exact'), d['parity'].get('bgr_equal'))"
  done
done
