run() { python bench.py --no-cpu-baseline "$@" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'], d['parity']['bgr_equal'], d['parity']['depth_bit_exact'])"; }
cp x_maps_amd/libxmaps_hip.so /tmp/keep.so
for t in 512 1024; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DXM_TILE_THREADS=$t x_maps_amd/csrc/xmaps_hip.hip -o x_maps_amd/libxmaps_hip.so
  echo "K1 threads $t"; run --slots 1; run --slots 4; run --slots 8; run --slots 16
done
cp /tmp/keep.so x_maps_amd/libxmaps_hip.so
