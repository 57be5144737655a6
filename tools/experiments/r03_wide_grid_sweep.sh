#!/bin/bash
# wide levels against the grid of a step launch (OCRGPU_STEP_GRID): is a level's time set by how its workgroups are scheduled?
for g in 512 1024 1536 2048; do
  OCRGPU_STEP_GRID=$g OCRGPU_ASYNC=0 timeout -k 10 200 python tools/bench_configs.py layered:10000000:100000 --no-parity 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('grid=$g', 'ms/resolve %.3f' % o['ms_per_resolve'], 'k_step ms %.3f' % o['ms_k_step'], 'sat/s %.3e' % o['sat_per_s'])"
done
