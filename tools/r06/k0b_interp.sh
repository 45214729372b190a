#!/bin/bash
# round 6: K0b's first round as four coarse probes + an interpolated dense window (variants/libxmaps_k0bi.so) against the 16-probe first round
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
( XM_LIB=variants/libxmaps_k0bi.so timeout 900 python -m pytest tests/test_gpu_cols.py tests/test_gpu_fused.py tests/test_gpu_shard_cols.py tests/test_gpu_sharded_cabi.py tests/test_gpu_sparse_groups.py -x -q -m gpu 2>&1 | tail -5
Q="--no-cpu-baseline --no-other-modes --no-host-path --no-pmc --no-other-configs --steps 40 --warmup 10"
for rep in 1 2 3; do for L in "" variants/libxmaps_k0bi.so; do
  XM_LIB=$L python bench.py $Q 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lib [$L] rep $rep: step ms', d.get('ms_per_step'), 'value', d.get('value'), d.get('roofline',{}).get('avg_launch_us'), 'depth exact', (d.get('parity') or {}).get('depth_bit_exact'), d.get('error'))"
done; done
XM_LIB=variants/libxmaps_k0bi.so timeout 600 python tools/fuzz_soak.py 5000 400 2>&1 | tail -3
) 2>&1 | tee gpurun_out/r06/k0b_interp.txt
