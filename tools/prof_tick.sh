#!/bin/bash
# kernel stats of the closed-loop tick: bash tools/prof_tick.sh <tag> <workload> [env assignments...]
# writes gpurun_out/<tag>_kernel_stats.csv
set -e -o pipefail
R=$PWD
TAG=$1; WL=$2; shift 2
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/tick_latency.py $WL 400 > $O/run.log 2>&1
cp $(find $O -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
tail -1 $O/run.log
rm -rf $O
