#!/bin/bash
# Experiment: eager pipelined loop with kernels switched off (-DXM_ABLATE build; skip_mask bit0=K0 bit1=K1 bit2=K2).
# Launch workers on (XM_WORKERS=1) so that the single-kernel rows are not capped by the host thread (2.5 us call + 2.7 us per launch).
set +e
cd "$(dirname "$0")/.."
cp x_maps_amd/libxmaps_hip.so /tmp/libxmaps_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_ABLATE x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
for m in 0 6 5 3 4 2 1; do
  echo -n "skip_mask=$m : "
  XM_WORKERS=1 XM_SKIP_MASK=$m python bench.py --no-cpu-baseline --no-parity --no-other-modes --steps 2000 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step', round(d['ms_per_step']*1e3,2), ' host enqueue', d['host_enqueue_us_per_step'])"
done
cp /tmp/libxmaps_hip.so.keep x_maps_amd/libxmaps_hip.so
