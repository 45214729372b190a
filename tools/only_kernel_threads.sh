#!/bin/bash
# Experiment: like only_kernel.sh but eager launches from several host threads (the host costs ~3 us per launch + ~2.5 us per
# ctypes call, which caps single-kernel runs from one thread).
set +e
cd "$(dirname "$0")/.."
cp x_maps_amd/libxmaps_hip.so /tmp/libxmaps_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_ABLATE x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
for m in 0 6 5 3 1; do
  echo "skip_mask=$m"
  XM_SKIP_MASK=$m python tools/two_threads_probe.py 2>&1 | grep threads
done
cp /tmp/libxmaps_hip.so.keep x_maps_amd/libxmaps_hip.so
