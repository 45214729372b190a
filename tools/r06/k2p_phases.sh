#!/bin/bash
# round 6: the pipelined K2's phase cycles on the ESL-like rig and at C-1M (ablation build: variants/libxmaps_abl.so, -DXM_ABLATE)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06
( for M in "--esl" ""; do XM_LIB=variants/libxmaps_abl.so timeout 300 python tools/k2p_phases.py $M 2>&1 | tail -12; done ) | tee gpurun_out/r06/k2p_phases.txt
