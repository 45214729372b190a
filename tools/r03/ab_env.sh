#!/bin/bash
# A/B of an environment switch in one session: `ab_env.sh VAR=VALUE [bench args]` -- bench lines alternating with / without it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
SW=$1; shift
mkdir -p gpurun_out/ab
Q="--no-cpu-baseline --no-other-modes --no-host-path"
for i in 1 2 3; do
  env $SW timeout 200 python bench.py $Q "$@" > gpurun_out/ab/var_$i.json 2> gpurun_out/ab/var_$i.err
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
