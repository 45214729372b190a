#!/bin/bash
# round 6: two K1 blocks per CU instead of three (XM_COLS_LDS_PAD: dynamic LDS beyond the carve-up), the room left to the persistent K2 -- does the overlap gain
# what K1 loses alone?
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 40 --warmup 10 --single-block"
( for rep in 1 2; do for V in "0 6" "8192 6" "8192 3" "8192 2" "34000 3" "34000 5"; do set -- $V
  python bench.py $Q --lib-option XM_COLS_LDS_PAD=$1 --lib-option XM_K2_PER_CU=$2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('K1 pad $1, K2 per CU $2, rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), 'exact', (d.get('parity') or {}).get('depth_bit_exact'), d.get('error'))"
done; done ) 2>&1 | tee gpurun_out/r06/k1_blocks.txt
