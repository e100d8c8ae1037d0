#!/bin/bash
# kernel stats of a bench workload: bash tools/prof_bench.sh <tag> <bench.py arguments...>
# writes gpurun_out/<tag>_kernel_stats.csv
set -e -o pipefail
R=$PWD
TAG=$1; shift 1
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py "$@" > $O/run.log 2>&1
cp $(find $O -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
tail -c 600 $O/run.log
rm -rf $O
