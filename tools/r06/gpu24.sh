#!/bin/bash
# round 6: the all-core C / OpenMP baseline in a child process without PyTorch's OpenMP runtime (median frame): five runs of both lines' CPU legs
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
for rep in 1 2 3 4 5; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-other-modes --no-host-path --no-pmc --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('C-1M rep $rep: numpy 1 core', c['value'], 'openmp', {k: c['all_cores_c_openmp'].get(k) for k in ('value','mean','best','cores','error')})"
  timeout 300 python bench.py --esl --steps 5 --warmup 2 --no-other-modes --no-host-path --no-pmc --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cpu_baseline']
print('ESL  rep $rep: numpy 1 core', c['value'], 'openmp', {k: c['all_cores_c_openmp'].get(k) for k in ('value','mean','best','cores','error')})"
done 2>&1 | tee gpurun_out/r06/openmp_baseline.txt
