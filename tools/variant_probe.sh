#!/bin/bash
# Experiment: compile-time variants of the kernels (block shapes, LDS budgets) -> pipelined throughput and kernel times.
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   slots',d['config']['frames_in_flight'],d['value'],round(d['ms_per_step']*1e3,2),d['roofline']['avg_launch_us'], d['parity']['depth_bit_exact'], d['parity']['bgr_equal'])"; }
cp x_maps_amd/libxmaps_hip.so /tmp/keep.so
while read -r flags; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $flags x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so || continue
  echo "flags: [$flags]"; run; run --slots 1; run --assume-sorted
done <<VARIANTS
${VARIANTS:-
-DXM_TILE_THREADS=512}
VARIANTS
cp /tmp/keep.so x_maps_amd/libxmaps_hip.so
