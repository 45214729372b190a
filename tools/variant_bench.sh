#!/bin/bash
# A/B of an alternative build of the library (XM_LIB) against the default one, same session: bash tools/variant_bench.sh <lib.so> [bench flags]
LIBV=$1; shift
for i in 1 2; do
  XM_LIB=$LIBV timeout 100 python bench.py --no-cpu-baseline --no-other-modes --no-host-path "$@" > gpurun_out/vb_var_$i.json 2>/dev/null
  timeout 100 python bench.py --no-cpu-baseline --no-other-modes --no-host-path "$@" > gpurun_out/vb_def_$i.json 2>/dev/null
done
