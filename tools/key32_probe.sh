for i in 1 2; do
XM_LIB=x_maps_amd/libxmaps_hip_key32.so timeout 200 python bench.py --no-cpu-baseline --no-parity --no-host-path > gpurun_out/r2g_key32_$i.json 2>gpurun_out/r2g_key32_$i.err
timeout 200 python bench.py --no-cpu-baseline --no-host-path > gpurun_out/r2g_key64_$i.json 2>gpurun_out/r2g_key64_$i.err
done
