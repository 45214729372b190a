#!/bin/bash
# Experiment: pipelined throughput of each kernel alone and in pairs (8 slots, hipGraph replay so that the host's ~3 us per
# launch does not cap the single-kernel runs), to see how the kernels' costs add up.  skip_mask bit0=K0 bit1=K1 bit2=K2.
set +e
cd "$(dirname "$0")/.."
cp x_maps_amd/libxmaps_hip.so /tmp/libxmaps_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_ABLATE x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
for m in 0 6 5 3 4 2 1; do
  for extra in "--graph --frames 32" "--graph --frames 32 --slots 16"; do
  echo -n "skip_mask=$m $extra : "
  XM_SKIP_MASK=$m python bench.py --no-cpu-baseline --no-parity --steps 400 $extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step', round(d['ms_per_step']*1e3,2))"
  done
done
cp /tmp/libxmaps_hip.so.keep x_maps_amd/libxmaps_hip.so
