cd $GRAFT_REPO_ROOT
cp x_maps_amd/libxmaps_hip.so /tmp/keep.so
for v in "30 100" "3 10"; do
  set -- $v
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_POLL_FIRST_US=$1 -DXM_POLL_NEXT_US=$2 x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
  echo "poll first $1 us next $2 us"
  for i in 1 2 3 4 5 6 7 8 9 10 11 12; do TRY=1 ORDER=upload-AB python tools/engine_order_probe.py 2>&1 | grep "pass 1 engine B" | cut -c28-75; done | sort | uniq -c | sort -rn | head -14
done
cp /tmp/keep.so x_maps_amd/libxmaps_hip.so
