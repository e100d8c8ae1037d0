#!/bin/bash
# kernel stats of the reception stage: bash tools/prof_events.sh <tag> <workload>
set -e -o pipefail
R=$PWD
TAG=$1; WL=$2
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/events_latency.py $WL 100 > $O/run.log 2>&1
cp $(find $O -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
tail -1 $O/run.log
rm -rf $O
