#!/bin/bash
# the other shapes of profiles/README.md on one MI355X: bash tools/round_shapes.sh > gpurun_out/shapes.log
R=$PWD
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = d.get('roofline', {}).get('stages', {})
sq = d.get('sequential_ticks') or {}
print('$1', 'us/tick %.2f' % (d['ms_per_tick'] * 1e3), 'links/s %.3e' % d['value'], 'seq us/tick %.1f' % (sq.get('ms_per_tick', 0) * 1e3), {k: round(v['us'], 1) for k, v in st.items()})"; }
B="python $R/bench.py --no-cpu-baseline --no-scale-probe --no-host-transfer"
$B --workload c4 --batch 16 2>/dev/null | line "c4 batch16"
$B --workload c4 --batch 1 --inflight 1 --steps 100 --warmup 10 2>/dev/null | line "c4 one tick at a time"
$B --workload c5 --steps 40 --warmup 12 2>/dev/null | line "c5"
$B --workload m1 --batch 16 2>/dev/null | line "m1 batch16"
$B --workload udgm 2>/dev/null | line "udgm"
$B --workload udgm_lossy 2>/dev/null | line "udgm_lossy"
for wl in c2 c3 udgm m1 udgm_lossy; do echo "tick_latency $wl: $(python $R/tools/tick_latency.py $wl 400 2>/dev/null | tail -1)"; done
echo "transmit_latency: $(python $R/tools/transmit_latency.py 2>/dev/null | tail -3 | tr '\n' ' ')"
