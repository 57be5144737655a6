#!/bin/bash
# A/B of the dense value array (OCRGPU_DENSE_VAL=0 / 1) on one box, interleaved: bash tools/ab_dense.sh
set -o pipefail
for rep in 1 2; do
  for dv in 0 1; do
    for spec in layered:10000000:100000 layered:10000000:10000 stencil:2048:4; do
      OCRGPU_DENSE_VAL=$dv OCRGPU_ASYNC=0 timeout -k 10 200 python tools/bench_configs.py $spec --no-parity 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('dense=$dv rep=$rep', o['workload'], 'ms/resolve %.3f' % o['ms_per_resolve'], 'k_step ms %.3f' % o['ms_k_step'], 'launches', o['launches'], 'sat/s %.3e' % o['sat_per_s'])" || exit 1
    done
  done
done
