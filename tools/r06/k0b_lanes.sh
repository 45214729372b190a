#!/bin/bash
# round 6: the pipelined C-1M step is bound by the HBM bytes of its three kernels (840 MB per group at 4.3 TB/s): variants that move fewer bytes --
# K0b with 16 lanes per boundary (variants/libxmaps_k0b16.so: -DXM_COLS_BOUNDS_LANES=16), K2 reading 2-byte pixel offsets (XM_K2_CONSEC=1)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 40 --warmup 10"
for rep in 1 2 3; do for V in "|" "variants/libxmaps_k0b16.so|" "|--lib-option XM_K2_CONSEC=1" "variants/libxmaps_k0b16.so|--lib-option XM_K2_CONSEC=1"; do
  L=${V%%|*}; OPT=${V##*|}
  XM_LIB=$L python bench.py $Q $OPT 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] [$OPT] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), 'depth exact', (d.get('parity') or {}).get('depth_bit_exact'), d.get('error'))"
done; done 2>&1 | tee gpurun_out/r06/k0b_lanes.txt
