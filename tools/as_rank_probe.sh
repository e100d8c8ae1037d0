#!/bin/bash
# quick look: one rank's share at several batch sizes.  usage: tools/as_rank_probe.sh OUT.jsonl WORKLOAD R:W "128 256 512" [extra]
out=$1; wl=$2; rw=$3; batches=$4; shift 4
: > "$out"
for b in $batches; do
  RM_HOST_TIMING=1 python3 bench.py --workload $wl --as-rank $rw --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch $b --steps 60 --warmup 6 "$@" >> "$out" 2>> "$out.err" || exit 1
done
