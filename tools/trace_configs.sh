#!/bin/bash
# kernel trace of one bench_configs workload: bash tools/trace_configs.sh <tag> <spec>
set -e -o pipefail
tag=$1; spec=$2
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o trace -- python3 $root/tools/bench_configs.py $spec --no-parity > $out/run.json 2> $out/run.err
