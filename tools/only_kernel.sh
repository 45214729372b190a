#!/bin/bash
# Experiment: pipelined throughput of each kernel alone (8 slots), to see what bounds the pipelined frame rate.
set -e
cd "$(dirname "$0")/.."
cp x_maps_amd/libxmaps_hip.so /tmp/libxmaps_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_ABLATE x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
for m in 0 6 5 3 4 2 1; do
  for sl in 1 8; do
  echo -n "skip_mask=$m slots=$sl : "
  XM_SKIP_MASK=$m python bench.py --no-cpu-baseline --no-parity --slots $sl --steps 400 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step', round(d['ms_per_step']*1e3,2))"
  done
done
cp /tmp/libxmaps_hip.so.keep x_maps_amd/libxmaps_hip.so
