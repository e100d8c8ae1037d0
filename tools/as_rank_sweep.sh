#!/bin/bash
# One GPU's view of a W-GPU run, for every rank R of W: bench.py --as-rank R:W (no collective) for the given workloads.
# usage: tools/as_rank_sweep.sh OUT.jsonl W "c3 c4" [extra bench args]
out=$1; W=$2; wls=$3; shift 3
: > "$out"
for wl in $wls; do
  python3 bench.py --workload $wl --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch 128 "$@" >> "$out" || exit 1
  for ((r=0; r<W; r++)); do
    python3 bench.py --workload $wl --as-rank $r:$W --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch 128 "$@" >> "$out" || exit 1
    echo "$wl rank $r done" >&2
  done
done
