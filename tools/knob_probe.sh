for i in 1 2; do
python3 bench.py --workload c3 --no-host-transfer --no-cpu-baseline --no-scale-probe --inflight 3 --batch 128 --steps 30 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('c3 128x3 steps 30', '| us/tick %.3f' % (d['ms_per_tick']*1e3), 'value %.3e' % d['value'], {k: round(v['us'],1) for k,v in st.items()})"
done
RM_FILTER_TICKS_PER_WG=2 python3 bench.py --workload c3 --no-host-transfer --no-cpu-baseline --no-scale-probe --inflight 3 --batch 128 --steps 30 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('c3 128x3 per_wg 2', '| us/tick %.3f' % (d['ms_per_tick']*1e3), 'value %.3e' % d['value'], {k: round(v['us'],1) for k,v in st.items()})"
