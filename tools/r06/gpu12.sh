#!/bin/bash
# round 6: owner-tile K1 with its block-uniform operands pinned in vector registers (fewer SGPR spills: v_readlane / v_writelane) -- tests, A/B
# (variants/libxmaps_novpin.so: the same hipcc line as x_maps_amd/_native.py's with -DXM_OWN_VPIN=0 -o variants/libxmaps_novpin.so)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_own.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/r06/t12.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t12.log; tail -4 gpurun_out/r06/t12.log
Q="--esl --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 20"
for rep in 1 2 3; do for L in variants/libxmaps_novpin.so ""; do
  XM_LIB=$L python bench.py $Q 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] rep $rep: K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), 'depth exact', (d.get('parity') or {}).get('depth_bit_exact'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/own_vpin_ab.txt
