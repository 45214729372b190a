#!/bin/bash
# round 6: K2 (pipelined) without the vmcnt(0) waits in its sample phase (the global table's look-up behind a branch of its own) -- tests, A/B on both rigs
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_k2pipe.py tests/test_gpu_configs.py tests/test_gpu_own.py tests/test_gpu_cols.py -q -m gpu -x > gpurun_out/r06/t13.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t13.log; tail -4 gpurun_out/r06/t13.log
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 20"
for rep in 1 2 3 4; do for L in variants/libxmaps_k2old.so ""; do for W in "--esl" ""; do
  XM_LIB=$L python bench.py $W $Q 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] [$W] rep $rep: K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), 'value', d.get('value'), 'depth exact', (d.get('parity') or {}).get('depth_bit_exact'), d.get('error'))"
done; done; done 2>&1 | tee gpurun_out/r06/k2_novmcnt_ab.txt
