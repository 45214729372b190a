#!/bin/bash
set -e
cd "$(dirname "$0")/.."
cp x_maps_amd/libxmaps_hip.so /tmp/libxmaps_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_BLOG x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
for s in ${SLOT_LIST:-8 4 1}; do SLOTS=$s python tools/block_timeline.py 2>&1 | grep -v amdgpu.ids; done
SLOTS=8 SORTED=1 python tools/block_timeline.py 2>&1 | grep -v amdgpu.ids
cp /tmp/libxmaps_hip.so.keep x_maps_amd/libxmaps_hip.so
