#!/bin/bash
set -e
cd "$(dirname "$0")/.."
cp x_maps_amd/libxmaps_hip.so /tmp/libxmaps_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_ABLATE x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
python tools/timeline.py 2>&1 | grep -v amdgpu.ids
cp /tmp/libxmaps_hip.so.keep x_maps_amd/libxmaps_hip.so
