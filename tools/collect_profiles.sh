#!/bin/bash
# Round evidence, run on the GPU box from the repository root:
#   RM_COMMIT=<commit> ROUND=r05 [PARTS="bench stats2 pmc_c3 pmc_m1 pmc_tick xcd pmc_c4 pmc_c5 pmc_dense ev asrank"] bash tools/collect_profiles.sh
# Writes under gpurun_out/$ROUND/; the summaries to be judged are then copied into profiles/.  One gpurun call may run
# 20 minutes: PARTS selects what a call collects (every part leaves its own summaries; pmc_traffic.json is per call and its
# entries are merged into profiles/pmc_traffic.json (python tools/pmc_merge.py gpurun_out/$ROUND/pmc_traffic_*.json)).
set -e -o pipefail
R=$PWD
ROUND=${ROUND:-r05}
PARTS=${PARTS:-bench stats2 pmc_c3 pmc_m1 pmc_tick pmc_c4 pmc_c5 pmc_dense ev asrank}
O=$R/gpurun_out/$ROUND
mkdir -p $O
stamp() { for f in "$@"; do echo "# commit ${RM_COMMIT:-unrecorded}" >> $f; done; }
has() { [[ " $PARTS " == *" $1 "* ]]; }
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-scale-probe --no-host-transfer"

# kernel stats of one bench command: stats <name> <bench args...>   -> ${ROUND}_<name>_kernel_stats.csv + ${ROUND}_<name>_bench.json
stats() {
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -- python3 $R/bench.py "$@" > $O/${ROUND}_${name}_bench.json 2> $O/st_$name.err
    cp $(find $O/st_$name -name "*kernel_stats.csv" | head -1) $O/${ROUND}_${name}_kernel_stats.csv
    stamp $O/${ROUND}_${name}_kernel_stats.csv
    (cd $R && python tools/trace_timed_avg.py $O/st_$name $O/${ROUND}_${name}_bench.json $O/${ROUND}_${name}_timed_kernel_avg.json > /dev/null) || echo "trace_timed_avg FAILED for $name (no ${ROUND}_${name}_timed_kernel_avg.json)" | tee -a $O/failures.log
    rm -rf $O/st_$name
    echo "stats $name done"
}
# counter passes of one bench command (separate passes: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2):
#   pmc <key> <ticks per launch> <bench args...>   -> ${ROUND}_<key>_pmc.csv, entry <key> of pmc_traffic.json
pmc() {
    local key=$1 tpl=$2; shift 2
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf_$key -- python3 $R/bench.py "$@" > $O/pf_$key.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw_$key -- python3 $R/bench.py "$@" > $O/pw_$key.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU --output-format csv -d $O/ps_$key -- python3 $R/bench.py "$@" > $O/ps_$key.log 2>&1
    (cd $R && python tools/pmc_traffic.py $O/pf_$key $O/pw_$key $key $tpl $O/${ROUND}_${key}_pmc.csv $O/pmc_traffic_${PARTS// /_}.json $O/ps_$key profiles/${ROUND}_${key}_pmc.csv)
    rm -rf $O/pf_$key $O/pw_$key $O/ps_$key
    echo "pmc $key done"
}

if has bench; then
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
cp $O/bench.json $O/${ROUND}_c3_bench_full.json
echo "bench done"
# the kernels' own intervals of the SAME command as the line's roofline (three contexts, 128 ticks per launch): what
# roofline.kernel_avg_us has to agree with
stats c3 $LEAN
stats c3_sequential $LEAN --inflight 1 --batch 1 --steps 400 --warmup 40
fi
if has stats2; then
stats c4 $LEAN --workload c4
stats c5 $LEAN --workload c5
stats m1 $LEAN --workload m1 --batch 16 --steps 24 --warmup 6
fi
if has pmc_c3; then
# the access-pattern calibration of FETCH_SIZE first (tools/pmc_traffic.py takes its factors from profiles/fetch_calibration.json)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -- $R/tools/fetch_calib > $O/calib.log 2>&1
(cd $R && python tools/fetch_calib.py $O/calib profiles/fetch_calibration.json > $O/fetch_calibration.log)
rm -rf $O/calib
pmc c3 128 $LEAN --inflight 1 --steps 6 --warmup 2
fi
if has pmc_m1; then pmc m1 16 $LEAN --workload m1 --inflight 1 --batch 16 --steps 12 --warmup 3; fi
if has pmc_tick; then pmc c3_tick 1 $LEAN --inflight 1 --batch 1 --steps 200 --warmup 20; fi
if has xcd; then
# the experiment of DESIGN.md section 4.7: the lone tick's frames dealt to the XCDs in eighths, node ids along a Morton curve
# (what the frame order has to be for it) -- time per tick without a profiler, then the counters, map off (0) and on (1, the default)
for v in 0 1; do
    RM_TICK_XCD_MAP=$v python3 $R/bench.py $LEAN --spatial-ids --inflight 1 --batch 1 --steps 400 --warmup 40 > $O/${ROUND}_c3_tick_spatial_xcd${v}_bench.json 2> $O/xcd$v.err || echo "xcd $v bench FAILED" | tee -a $O/failures.log
    RM_TICK_XCD_MAP=$v pmc c3_tick_spatial_xcd$v 1 $LEAN --spatial-ids --inflight 1 --batch 1 --steps 200 --warmup 20
done
fi
if has pmc_c4; then pmc c4 128 $LEAN --workload c4 --inflight 1 --steps 4 --warmup 2; fi
if has pmc_c5; then pmc c5 128 $LEAN --workload c5 --steps 6 --warmup 2; pmc c5_tick 1 $LEAN --workload c5 --batch 1 --steps 60 --warmup 12; fi
if has pmc_dense; then
stats dense --dense-only
pmc dense 1 --dense-only
fi
if has ev; then
# the reception stage (device events): tick + drain, deliveries to the host
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ev_stats -- python3 $R/tools/events_latency.py c3 100 > $O/ev_stats.log 2>&1
cp $(find $O/ev_stats -name "*kernel_stats.csv" | head -1) $O/${ROUND}_c3_events_kernel_stats.csv
grep -o "{\"workload\".*}" $O/ev_stats.log > $O/${ROUND}_c3_events.json || echo "events_latency printed no result line" | tee -a $O/failures.log
stamp $O/${ROUND}_c3_events_kernel_stats.csv
rm -rf $O/ev_stats
# ... and the closed loop through the C ABI
(cd $R && tools/loop_latency > $O/${ROUND}_closed_loop_c_abi.jsonl 2> $O/loop_latency.err) || echo "loop_latency FAILED (see loop_latency.err)" | tee -a $O/failures.log
# ... and what the PCIe link takes of a tick's 0.55 MB of delivery records, by store width (tools/pcie_store_width.hip)
if [ -x $R/tools/pcie_store_width ]; then (cd $R && tools/pcie_store_width > $O/${ROUND}_pcie_store_width.jsonl 2> /dev/null) || echo "pcie_store_width FAILED" | tee -a $O/failures.log; fi
echo "events done"
fi
# one rank's share of an 8-GPU run, rank by rank (compute side of strong scaling); asrank = all three, or asrank_c3 / _c4 / _c5
cd $R
for wl in c3 c4 c5; do
    if has asrank || has asrank_$wl; then
        (tools/as_rank_sweep.sh $O/${ROUND}_asrank8_$wl.jsonl 8 $wl 512 128 2> $O/asrank_$wl.err && python tools/as_rank_table.py $O/${ROUND}_asrank8_$wl.jsonl > $O/${ROUND}_asrank8_$wl.txt) || echo "asrank $wl FAILED (see asrank_$wl.err)" | tee -a $O/failures.log
    fi
done
if has server; then
# the TCP server's own time per step at the BASELINE frame size (tools/server_latency.py: nodes frames steps frame-bytes)
for th in 8 16; do
    RSIM_SERVER_THREADS=$th python3 tools/server_latency.py 100000 1000 12 127 > $O/${ROUND}_server_100k_1000_127B_t$th.txt 2>&1 || echo "server_latency t$th FAILED" | tee -a $O/failures.log
done
fi
echo "all done: $PARTS"
