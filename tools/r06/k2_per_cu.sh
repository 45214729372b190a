#!/bin/bash
# round 6: the pipelined K2's blocks per CU (XM_K2_PER_CU) under overlap -- four groups in flight -- and alone
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 40 --warmup 10"
( for rep in 1 2; do for P in ${PS:-6 5 4 3}; do for M in ${MS-"" "--esl"}; do
  python bench.py $Q $M --lib-option XM_K2_PER_CU=$P 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('per_cu $P [$M] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), d.get('error'))"
done; done; done ) 2>&1 | tee gpurun_out/r06/k2_per_cu.txt
