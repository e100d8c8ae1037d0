#!/bin/bash
# The randomized parity suites once per developer knob (forced filter variants, hipGraph replay, no
# shadowing table, no one-launch transmit, a new sort after every node change).  Run on the GPU box
# from the repository root:  bash tools/knob_sweep.sh [blocks]
B=${1:-24}
for knobs in "RM_FILTER=wg" "RM_FILTER=wg RM_WG_RPT=4" "RM_FILTER=wg RM_WG_RPT=2" "RM_FILTER=grid" "RM_GRAPH=1" \
             "RM_NO_SHADOW_TABLE=1" "RM_NO_ONE_LAUNCH=1" "RM_RESORT_AFTER=0" "RM_NO_REC32=1" "RM_EXACT_GRID=1" "RM_EXACT_GRID=7" "RM_EXACT_GRID=256"; do
    echo "== $knobs"
    env $knobs RM_STRESS_BLOCKS=$B timeout -k 10 900 python -m pytest tests/test_gpu_random_midsize.py \
        tests/test_gpu_random_stress.py tests/test_gpu_batch.py tests/test_gpu_api.py tests/test_gpu_sharded.py -q -x 2>&1 | tail -2
done
