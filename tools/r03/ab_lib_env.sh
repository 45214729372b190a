#!/bin/bash
# A/B of an alternative build together with environment switches, alternating with the default build in one session:
#   ab_lib_env.sh variants/libx.so "VAR=1 OTHER=2" [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
LIBV=$1; SW=$2; shift 2
mkdir -p gpurun_out/ab
Q="--no-cpu-baseline --no-other-modes --no-host-path"
for i in 1 2 3; do
  env XM_LIB=$LIBV $SW timeout 200 python bench.py $Q "$@" > gpurun_out/ab/var_$i.json 2> gpurun_out/ab/var_$i.err
  timeout 200 python bench.py $Q "$@" > gpurun_out/ab/def_$i.json 2> gpurun_out/ab/def_$i.err
done
for f in var_1 def_1 var_2 def_2 var_3 def_3; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/ab/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
    r=d['roofline']; ks=r.get('kernels',{})
    print(sys.argv[1], d['value'], r['frac'], {k:v.get('us_per_frame') for k,v in ks.items()}, d.get('parity',{}).get('depth_bit_exact'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
