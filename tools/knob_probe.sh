for inf in 1 3; do
  python3 bench.py --workload c3 --as-rank 0:8 --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight $inf --batch 512 --steps 40 --warmup 6 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('rank 0:8 inflight $inf', '| us/tick %.3f' % (d['ms_per_tick']*1e3), {k: round(v['us'],1) for k,v in st.items()})"
done
for cfg in "64 2" "128 3"; do
  set -- $cfg
  python3 bench.py --workload c3 --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight $2 --batch $1 --steps $((3840/$1)) --warmup $((384/$1)) 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('whole batch $1 ctx $2', '| us/tick %.3f' % (d['ms_per_tick']*1e3), 'value %.3e' % d['value'], {k: round(v['us'],1) for k,v in st.items()})"
done
python3 bench.py --workload c4 --as-rank 0:8 --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch 256 --steps 20 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('c4 rank 0:8', '| us/tick %.3f' % (d['ms_per_tick']*1e3), {k: round(v['us'],1) for k,v in st.items()})"
