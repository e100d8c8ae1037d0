#!/bin/bash
# Round evidence, run on the GPU box from the repository root:
#   RM_COMMIT=<commit> ROUND=r03 [PARTS="bench pmc m1 tick c5ev asrank"] bash tools/collect_profiles.sh
# Writes under gpurun_out/$ROUND/; the summaries to be judged are then copied into profiles/.  One gpurun call may run
# 20 minutes: PARTS selects what a call collects (every part leaves its own summaries; pmc_traffic.json is per call and its
# entries are merged into profiles/pmc_traffic.json by hand).
set -e -o pipefail
R=$PWD
ROUND=${ROUND:-r03}
PARTS=${PARTS:-bench pmc m1 tick c5ev asrank}
O=$R/gpurun_out/$ROUND
mkdir -p $O
stamp() { for f in "$@"; do echo "# commit ${RM_COMMIT:-unrecorded}" >> $f; done; }
has() { [[ " $PARTS " == *" $1 "* ]]; }
cd /tmp && export TMPDIR=/tmp
if has bench; then
python $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-scale-probe > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seq -- python $R/bench.py --no-cpu-baseline --no-scale-probe --inflight 1 --batch 1 --steps 400 --warmup 40 > $O/stats_seq.log 2>&1
echo "sequential stats done"
cp $O/bench.json $O/${ROUND}_c3_bench.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${ROUND}_c3_kernel_stats.csv
cp $(find $O/stats_seq -name "*kernel_stats.csv" | head -1) $O/${ROUND}_c3_sequential_kernel_stats.csv
stamp $O/${ROUND}_c3_kernel_stats.csv $O/${ROUND}_c3_sequential_kernel_stats.csv
rm -rf $O/stats $O/stats_seq
fi
if has pmc; then
# the access-pattern calibration of FETCH_SIZE first (tools/pmc_traffic.py takes its factors from profiles/fetch_calibration.json)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -- $R/tools/fetch_calib > $O/calib.log 2>&1
(cd $R && python tools/fetch_calib.py $O/calib profiles/fetch_calibration.json > $O/fetch_calibration.log)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --no-cpu-baseline --no-scale-probe --no-host-transfer --inflight 1 --steps 6 --warmup 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --no-cpu-baseline --no-scale-probe --no-host-transfer --inflight 1 --steps 6 --warmup 2 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- python $R/bench.py --no-cpu-baseline --no-scale-probe --no-host-transfer --inflight 1 --steps 6 --warmup 2 > $O/pmc_sq.log 2>&1
# ... and the same vector-issue counters with the bench's own number of contexts sharing the device (the counts must agree:
# instructions do not depend on who else runs)
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq3 -- python $R/bench.py --no-cpu-baseline --no-scale-probe --no-host-transfer --steps 9 --warmup 3 > $O/pmc_sq3.log 2>&1
echo "pmc c3 done"
(cd $R && python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write c3 128 $O/${ROUND}_c3_pmc.csv $O/pmc_traffic.json $O/pmc_sq)
(cd $R && python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write c3_three_contexts 128 $O/${ROUND}_c3_pmc_three_contexts.csv $O/pmc_traffic.json $O/pmc_sq3)
rm -rf $O/calib $O/pmc_sq3 $O/pmc_fetch $O/pmc_write $O/pmc_sq
fi
if has m1; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/m1_stats -- python $R/bench.py --no-cpu-baseline --no-scale-probe --workload m1 --batch 16 --steps 24 --warmup 6 > $O/m1_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/m1_fetch -- python $R/bench.py --no-cpu-baseline --no-scale-probe --workload m1 --inflight 1 --batch 16 --steps 12 --warmup 3 > $O/m1_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/m1_write -- python $R/bench.py --no-cpu-baseline --no-scale-probe --workload m1 --inflight 1 --batch 16 --steps 12 --warmup 3 > $O/m1_write.log 2>&1
echo "m1 done"
(cd $R && python tools/pmc_traffic.py $O/m1_fetch $O/m1_write m1 16 $O/${ROUND}_m1_pmc.csv $O/pmc_traffic.json)
cp $(find $O/m1_stats -name "*kernel_stats.csv" | head -1) $O/${ROUND}_m1_kernel_stats.csv
stamp $O/${ROUND}_m1_kernel_stats.csv
rm -rf $O/m1_stats $O/m1_fetch $O/m1_write
fi
if has tick; then
# the closed-loop tick (one launch per tick): its own PMC passes, keyed by the workload name c3_tick
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tick_fetch -- python $R/bench.py --no-cpu-baseline --no-scale-probe --no-host-transfer --inflight 1 --batch 1 --steps 200 --warmup 20 > $O/tick_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tick_write -- python $R/bench.py --no-cpu-baseline --no-scale-probe --no-host-transfer --inflight 1 --batch 1 --steps 200 --warmup 20 > $O/tick_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/tick_sq -- python $R/bench.py --no-cpu-baseline --no-scale-probe --no-host-transfer --inflight 1 --batch 1 --steps 200 --warmup 20 > $O/tick_sq.log 2>&1
echo "tick pmc done"
(cd $R && python tools/pmc_traffic.py $O/tick_fetch $O/tick_write c3_tick 1 $O/${ROUND}_c3_tick_pmc.csv $O/pmc_traffic.json $O/tick_sq)
rm -rf $O/tick_fetch $O/tick_write $O/tick_sq
fi
if has c5ev; then
# configs[4]: the SINR medium with frames that stay on the air (the tick by scan, rm_airscan.hip)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python $R/bench.py --no-cpu-baseline --no-host-transfer --workload c5 --steps 40 --warmup 12 > $O/c5_stats.log 2>&1
# the reception stage (device events): tick + drain, deliveries to the host
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ev_stats -- python $R/tools/events_latency.py c3 100 > $O/ev_stats.log 2>&1
echo "c5 / events done"
cp $(find $O/c5_stats -name "*kernel_stats.csv" | head -1) $O/${ROUND}_c5_kernel_stats.csv
cp $(find $O/ev_stats -name "*kernel_stats.csv" | head -1) $O/${ROUND}_c3_events_kernel_stats.csv
grep -h '"metric"' $O/c5_stats.log > $O/${ROUND}_c5_bench.json || true
grep -o "{\"workload\".*}" $O/ev_stats.log > $O/${ROUND}_c3_events.json || true
stamp $O/${ROUND}_c5_kernel_stats.csv $O/${ROUND}_c3_events_kernel_stats.csv
# ... and the closed loop through the C ABI
(cd $R && tools/loop_latency > $O/${ROUND}_closed_loop_c_abi.jsonl 2> /dev/null || true)
rm -rf $O/c5_stats $O/ev_stats
fi
if has asrank; then
# one rank's share of an 8-GPU run, rank by rank (compute side of strong scaling)
cd $R
tools/as_rank_sweep.sh $O/${ROUND}_asrank8_c3.jsonl 8 c3 512 128 2> /dev/null && python tools/as_rank_table.py $O/${ROUND}_asrank8_c3.jsonl > $O/${ROUND}_asrank8_c3.txt
tools/as_rank_sweep.sh $O/${ROUND}_asrank8_c4.jsonl 8 c4 256 32 2> /dev/null && python tools/as_rank_table.py $O/${ROUND}_asrank8_c4.jsonl > $O/${ROUND}_asrank8_c4.txt
fi
echo "all done: $PARTS"
