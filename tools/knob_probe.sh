for kn in "X=1" "RM_EXACT_GRID=2" "RM_EXACT_GRID=1" "RM_EXACT_GRID=2 RM_BATCH_SHARDS=16" "RM_WG_RPT=2"; do
env $kn python3 bench.py --workload c3 --as-rank 0:8 --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch 512 --steps 24 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('c3 rank 0:8 $kn', '| us/tick %.3f' % (d['ms_per_tick']*1e3), {k: round(v['us'],1) for k,v in st.items()})"
env $kn python3 bench.py --workload c4 --as-rank 0:8 --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch 256 --steps 20 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('c4 rank 0:8 $kn', '| us/tick %.3f' % (d['ms_per_tick']*1e3), {k: round(v['us'],1) for k,v in st.items()})"
done
