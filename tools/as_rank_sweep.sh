#!/bin/bash
# One GPU's view of a W-GPU run, for every rank R of W: bench.py --as-rank R:W (no collective; the frames arrive in the
# layout the all-gather leaves them in) next to the whole tick on one GPU with the same settings.
# usage: tools/as_rank_sweep.sh OUT.jsonl W WORKLOAD BATCH WHOLE_BATCH [extra bench args]
# (WHOLE_BATCH: ticks per launch of the one-GPU run -- its result slots are sized for all the links of a tick)
out=$1; W=$2; wl=$3; b=$4; wb=$5; shift 5
: > "$out"
python3 bench.py --workload $wl --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch $wb --steps 24 --warmup 8 "$@" >> "$out" || exit 1
for ((r=0; r<W; r++)); do
  python3 bench.py --workload $wl --as-rank $r:$W --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch $b --steps 24 --warmup 8 "$@" >> "$out" || exit 1
  echo "$wl rank $r done" >&2
done
