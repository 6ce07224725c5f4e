#!/bin/bash
mkdir -p gpurun_out/r04b
for v in default "$@"; do
  if [ "$v" = default ]; then unset BTRAPZ_HIP_LIB; else export BTRAPZ_HIP_LIB=$PWD/scratch/variants/$v/libbtrapz_hip.so; fi
  echo "== $v"
  AB_CASES=0,1,2,3 python tools/ab_variants.py cur=${BTRAPZ_HIP_LIB:-default} | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('cur {'):
        d = json.loads(l.split(' ', 1)[1])
        print({k: (v['ms'], round(v['it'] + 1, 3), v['itmax'], v['solved']) for k, v in d.items() if k.endswith('two')})
"
  python tools/jittered_sets_iterations.py 2>/dev/null | head -1
done
