#!/bin/bash
# round 6, fourth GPU call: K2's direct 12-byte BGR stores (tests + A/B against the staged rows), the on-arrival test, the ubench's band lines
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_on_arrival.py tests/test_gpu_k2pipe.py tests/test_gpu_own.py tests/test_gpu_fused.py tests/test_gpu_a4_bruteforce.py tests/test_gpu_cols.py -q -m gpu -x > gpurun_out/r06/t4.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t4.log; tail -6 gpurun_out/r06/t4.log
Q="--esl --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 20"
for rep in 1 2; do for L in "" variants/libxmaps_staged.so; do
  XM_LIB=$L python bench.py $Q 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] rep $rep: K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), 'Gev/s', d.get('value'), 'bgr-only', (d.get('other_modes') or {}).get('groups_bgr_only',{}).get('ms_per_step'), d.get('parity',{}).get('bgr_equal'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/k2_direct_ab.txt
for L in "" variants/libxmaps_staged.so; do
  XM_LIB=$L python bench.py --no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C-1M lib [$L]: K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), 'Gev/s', d.get('value'), d.get('error'))"
done 2>&1 | tee -a gpurun_out/r06/k2_direct_ab.txt
tools/ubench/tile_stream 2>&1 | grep -i "barrier\|bands" | tee gpurun_out/r06/tile_stream_bands.txt
