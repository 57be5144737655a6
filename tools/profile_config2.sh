#!/bin/bash
# Round profile of the default bench workload (config 2): kernel trace + two PMC passes, each its own run.
# usage (on the GPU box, from the repo root): bash tools/profile_config2.sh <tag>
set -e -o pipefail
tag=${1:-r03b}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 170 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $root/bench.py > $out/trace.json 2> $out/trace.err
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_fetch.json 2> $out/pmc_fetch.err
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o pmc -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_write.json 2> $out/pmc_write.err
timeout -s KILL 170 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $out/pmc_atom -o pmc -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_atom.json 2> $out/pmc_atom.err
ls -R $out | head -50
