for kn in "X=1" "RM_FPW=2" "RM_FPW=4" "RM_FPW=8" "RM_BATCH_SHARDS=64" "RM_EXACT_GRID=8"; do
env $kn python3 bench.py --workload c3 --as-rank 0:2 --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch 128 --steps 30 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('c3 rank 0:2 $kn', '| us/tick %.3f' % (d['ms_per_tick']*1e3), {k: round(v['us'],1) for k,v in st.items()})"
done
for kn in "X=1" "RM_FPW=4" "RM_FPW=8" "RM_FPW=16"; do
env $kn python3 bench.py --workload c3 --as-rank 0:4 --no-host-transfer --no-scale-probe --no-cpu-baseline --inflight 3 --batch 256 --steps 30 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages']
print('c3 rank 0:4 $kn', '| us/tick %.3f' % (d['ms_per_tick']*1e3), {k: round(v['us'],1) for k,v in st.items()})"
done
