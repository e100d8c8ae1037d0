#!/bin/bash
# The randomized parity suites once per developer knob (forced filter variants, hipGraph replay, no
# shadowing table, no one-launch transmit, a new sort after every node change).  Run on the GPU box
# from the repository root:  bash tools/knob_sweep.sh [blocks] [first knob] [last knob]
# Round 2 added: the three-launch sweep instead of the one-launch tick (RM_FRAME_TICK=0), the SINR lists rebuilt every
# tick (RM_AIR_LISTS=0), the tiled filter instead of the per-frame candidate kernel (RM_FRAMES_CAND=0), and the SINR /
# reception-stage / group suites in every run.
# Round 3 added: several ticks per filter workgroup on tables that would take one (RM_FILTER_TICKS_PER_WG=3), the copied
# instead of the zero-copy records of a flushed tick (RM_NO_ZERO_COPY=1), few / very many frames per reorder wave (RM_FPW),
# another SINR grid (RM_SINR_GX=5), groups without RCCL (RM_GROUP_NO_RCCL=1), the SINR medium's lone ticks with the per-receiver
# lists instead of by scan (RM_SINR_SCAN=0; RM_AIR_LISTS=0 only means something with it), and the comm suite in every run.
# Round 4 added: the dense tick wherever the configuration allows it / never (RM_DENSE_TICK), the near-frame lists of the batch
# filter at any size (RM_NEAR_LISTS=2 with the 1024-receiver workgroups), the reception stage with the tick's append on its own
# (RM_EV_FUSE=0) and with all / none of the deliveries written by the emit kernel (RM_EV_SHARE), a pair list that starts too small
# (RM_OV_PAIR_CAP), and the overlap and dense suites in every run.
# Round 5 added: a partitioned batch without the rank's frame list (RM_RANK_FRAMES=0), the SINR batches with per-link entries instead
# of the per-receiver sums (RM_SINR_ACC=0), the dense tick that writes its records at once (RM_DENSE_LAZY=0), host views with the
# rssi per link instead of per packet (RM_HOST_LINK_RSSI=1), the batch filter that takes the near-frame lists of sixteen ticks at
# a time on tables and batches of any size (RM_NEAR_LISTS=2 RM_WG_RPT=4 RM_FILTER_TICKS_PER_WG=5 / 40 RM_FILTER_GROUP=1) and never (RM_FILTER_GROUP=0), the lone tick's
# frames dealt to the XCDs in turn instead of in eighths (RM_TICK_XCD_MAP=0), the drain's rank pass from global memory (RM_EV_EMIT_LDS=0),
# the reorder stage's waves with single frames / runs of 64 / of 8 consecutive frames (RM_REORDER_RUN).
B=${1:-24}
FIRST=${2:-0}
LAST=${3:-99}
K=("RM_FILTER=wg" "RM_FILTER=wg RM_WG_RPT=4" "RM_FILTER=wg RM_WG_RPT=2" "RM_FILTER=grid" "RM_GRAPH=1"
   "RM_NO_SHADOW_TABLE=1" "RM_NO_ONE_LAUNCH=1" "RM_RESORT_AFTER=0" "RM_NO_REC32=1" "RM_EXACT_GRID=1" "RM_EXACT_GRID=7" "RM_EXACT_GRID=256"
   "RM_FRAME_TICK=0" "RM_SINR_SCAN=0 RM_AIR_LISTS=0" "RM_FILTER=wg RM_FRAMES_CAND=0" "RM_FR_FLAT_MAX=0" "RM_FR_NO_SHADOW=1" "RM_SINR_FRAMES=0" "RM_SINR_FRAMES=0 RM_FILTER=wg"
   "RM_FILTER_TICKS_PER_WG=3" "RM_NO_ZERO_COPY=1" "RM_FPW=3" "RM_FPW=200" "RM_SINR_GX=5" "RM_GROUP_NO_RCCL=1" "RM_SINR_SCAN=0"
   "RM_DENSE_TICK=1" "RM_DENSE_TICK=0" "RM_NEAR_LISTS=2 RM_WG_RPT=4" "RM_EV_FUSE=0" "RM_EV_SHARE=0" "RM_EV_SHARE=1" "RM_OV_PAIR_CAP=4096"
   "RM_RANK_FRAMES=0" "RM_SINR_ACC=0" "RM_DENSE_LAZY=0" "RM_HOST_LINK_RSSI=1"
   "RM_NEAR_LISTS=2 RM_WG_RPT=4 RM_FILTER_TICKS_PER_WG=5 RM_FILTER_GROUP=1" "RM_NEAR_LISTS=2 RM_WG_RPT=4 RM_FILTER_TICKS_PER_WG=40 RM_FILTER_GROUP=1" "RM_FILTER_GROUP=0" "RM_TICK_XCD_MAP=0" "RM_EV_EMIT_LDS=0" "RM_REORDER_RUN=0" "RM_REORDER_RUN=6" "RM_REORDER_RUN=3 RM_FPW=5")
for i in "${!K[@]}"; do
    if [ $i -lt $FIRST ] || [ $i -gt $LAST ]; then continue; fi
    knobs="${K[$i]}"
    echo "== $knobs"
    env $knobs RM_STRESS_BLOCKS=$B timeout -k 10 900 python -m pytest tests/test_gpu_random_midsize.py \
        tests/test_gpu_random_stress.py tests/test_gpu_batch.py tests/test_gpu_api.py tests/test_gpu_sharded.py tests/test_gpu_logdist.py \
        tests/test_gpu_events.py tests/test_gpu_group.py tests/test_gpu_parity.py tests/test_gpu_comm.py tests/test_gpu_overlap.py tests/test_gpu_dense.py -q -x 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -12
done
