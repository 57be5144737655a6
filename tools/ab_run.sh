#!/bin/bash
# A/B on one box: bash tools/ab_run.sh <name>=<lib.so> ...   (each twice, interleaved; prints the bench lines)
set -o pipefail
names=()
for rep in 1 2; do
  for spec in "$@"; do
    n=${spec%%=*}; lib=${spec#*=}
    OCRGPU_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab_${n}$rep.json 2> gpurun_out/ab_${n}$rep.err || exit 1
    names+=("${n}$rep")
  done
done
python tools/ab_print.py "${names[@]}"
