#!/bin/bash
# Experiment only (results are wrong by construction): which part of K1 limits the PIPELINED frame rate?
cd "$(dirname "$0")/.."
cp x_maps_amd/libxmaps_hip.so /tmp/libxmaps_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_ABLATE x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
for a in 0 1 2 4 8 5; do
  for extra in "" "--assume-sorted"; do
  echo -n "XM_ABLATE=$a $extra : "
  XM_ABLATE=$a python bench.py --no-cpu-baseline --no-parity --slots 8 --steps 400 $extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step', round(d['ms_per_step']*1e3,2), d['roofline']['avg_launch_us'])"
  done
done
cp /tmp/libxmaps_hip.so.keep x_maps_amd/libxmaps_hip.so
