#!/bin/bash
# round 6: the activity rule's variants (own pixel, strict comparison) as configuration -- tests; ingest soak
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_activity.py tests/test_gpu_ingest.py tests/test_gpu_configs.py tests/test_gpu_api.py -q -m gpu -x > gpurun_out/r06/t17.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06/t17.log; tail -6 gpurun_out/r06/t17.log
timeout 900 python tools/ingest_soak.py 500 1500 > gpurun_out/r06/soak17.log 2>&1; tail -3 gpurun_out/r06/soak17.log
timeout 900 python tools/fuzz_soak.py 420 1500 > gpurun_out/r06/fuzz17.log 2>&1; tail -2 gpurun_out/r06/fuzz17.log
