#!/bin/bash
# where does a tick of the bench's tick_events leg spend its time?  kernel timeline of the leg (rocprofv3 --kernel-trace)
R=$PWD
O=$R/gpurun_out/prof_evgaps; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/bench.py --no-cpu-baseline --no-scale-probe --steps 4 --warmup 1 > $O/run.log 2>&1
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void rm::", "").replace("rm::", "")[:20] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_ev_finish")]
# one drain in the middle of the leg: print the timeline between two k_ev_finish
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]["End_Timestamp"])
for i in range(a + 1, b + 1):
    print("%-22s start +%7.1f us  dur %6.1f us  queue %s" % (names[i], (int(rows[i]["Start_Timestamp"]) - t0) / 1e3,
          (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, rows[i].get("Queue_Id")))
PY
rm -rf $O
