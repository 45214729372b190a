#!/bin/bash
# Experiment: like timeline.sh, but one row per wave of block 0 (shows how far apart the 16 waves of a block run).
set -e
cd "$(dirname "$0")/.."
cp x_maps_amd/libxmaps_hip.so /tmp/libxmaps_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_ABLATE -DXM_STAMP_WAVES x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
XM_STAMP_WAVES=1 python tools/timeline.py 2>&1 | grep -v amdgpu.ids | tail -18
cp /tmp/libxmaps_hip.so.keep x_maps_amd/libxmaps_hip.so
