#!/bin/bash
# round 6: K2's loader in whole-column passes (one division per item) + the metadata of the item after next fetched at the top of the iteration
# (the variant libraries are builds of tools/r06/patches/k2_branch_free.patch applied to xmaps_k2pipe.hpp: its first half = libxmaps_k2m.so, all of it = libxmaps_k2bf.so;
#  variants/libxmaps_base.so = a build of the commit itself: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared x_maps_amd/csrc/xmaps_hip.hip -o <lib>)
# (variants/libxmaps_k2m.so) against the product library, C-1M and ESL-like, pipelined and one group at a time
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
( XM_LIB=variants/libxmaps_k2m.so timeout 900 python -m pytest tests/test_gpu_k2pipe.py tests/test_gpu_fused.py tests/test_gpu_configs.py tests/test_gpu_own.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
XM_LIB=variants/libxmaps_k2m.so timeout 600 python tools/fuzz_soak.py 7000 150 2>&1 | tail -1
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 40 --warmup 10"
for rep in 1 2 3; do for L in "" variants/libxmaps_k2m.so; do for M in "" "--groups-in-flight 1" "--esl" "--esl --groups-in-flight 1"; do
  XM_LIB=$L python bench.py $Q $M 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] [$M] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), 'exact', (d.get('parity') or {}).get('depth_bit_exact', (d.get('parity') or {}).get('group_last_frame_depth_bit_exact')), d.get('error'))"
done; done; done ) 2>&1 | tee gpurun_out/r06/k2_meta.txt
