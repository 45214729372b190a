#!/bin/bash
# round 6: soaks of the round's final build
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
timeout 1200 python tools/fuzz_soak.py 2000 3000 2>&1 | tail -2 | tee gpurun_out/r06/soak_fuzz.txt
timeout 1200 python tools/ingest_soak.py 4000 3000 2>&1 | tail -2 | tee gpurun_out/r06/soak_ingest.txt
timeout 900 python tools/evt3_soak.py 0 1000 2>&1 | tail -2 | tee gpurun_out/r06/soak_evt3.txt
