#!/bin/bash
mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --no-parity --steps 40 --warmup 10"
for rep in 1 2 3; do for OPT in "" "--lib-option XM_SKIP_K0B=8"; do
  XM_LIB=variants/libxmaps_skipk0b.so python bench.py $Q $OPT 2>>gpurun_out/r06/skip_err.txt | tail -1 | tee -a gpurun_out/r06/skip_raw.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$OPT] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/skip_k0b.txt
