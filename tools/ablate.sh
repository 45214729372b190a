#!/bin/bash
# Experiment only: build libxmaps_hip.so with the ablation hooks (-DXM_ABLATE) and time K1 with pieces switched off.
# Results are WRONG by construction (parity gate is bypassed with --no-parity); never ship this build.
set -e
cd "$(dirname "$0")/.."
cp x_maps_amd/libxmaps_hip.so /tmp/libxmaps_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_ABLATE x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
for a in 0 1 2 4 8 3 7 15; do
  echo -n "XM_ABLATE=$a  "
  XM_ABLATE=$a python bench.py --no-cpu-baseline --no-parity --slots 1 --steps 200 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_us'])"
done
cp /tmp/libxmaps_hip.so.keep x_maps_amd/libxmaps_hip.so
