#!/bin/bash
# round 6: K2 (pipelined) with ONE 16-byte patch load per thread on rigs whose patches are <= 256 quads (59 VGPRs: seven / eight blocks per CU instead of six) -- tests, A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_k2pipe.py tests/test_gpu_configs.py tests/test_gpu_own.py -q -m gpu -x > gpurun_out/r06/t15.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t15.log; tail -4 gpurun_out/r06/t15.log
Q="--esl --no-cpu-baseline --no-host-path --no-pmc --no-other-configs --steps 20"
for rep in 1 2 3 4; do for OPT in "XM_K2_UN1=0" "XM_K2_UN1=1" "XM_K2_NLDS_MAX=1700" "XM_K2_NLDS_MAX=1400" "XM_K2_PER_CU=6"; do
  python bench.py $Q --lib-option $OPT 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$OPT] rep $rep: K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), 'bgr-only', (d.get('other_modes') or {}).get('groups_bgr_only',{}).get('ms_per_step'), 'depth exact', (d.get('parity') or {}).get('depth_bit_exact'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/k2_un1_ab.txt
# C-1M: the two-pixels-per-thread kernel compiled for seven / eight waves per SIMD (72 / 64 VGPRs) against six (78)
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 20"
for rep in 1 2 3; do for L in "" variants/libxmaps_k2w7.so variants/libxmaps_k2w8.so; do
  XM_LIB=$L python bench.py $Q 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] rep $rep: K0b/K1/K2 us', d.get('roofline',{}).get('avg_launch_us'), 'step ms', d.get('ms_per_step'), 'value', d.get('value'), 'depth exact', (d.get('parity') or {}).get('depth_bit_exact'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/k2_waves_ab.txt
