#!/bin/bash
# c5 kernel durations (rocprofv3 kernel trace) under a few knobs: bash tools/c5_knobs.sh "K=V ..." ...
R=$PWD
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  echo "== $kv"
  O=$R/gpurun_out/prof_knob; rm -rf $O; mkdir -p $O
  env $kv rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --workload c5 --steps 40 --warmup 12 --no-host-transfer --no-cpu-baseline > $O/run.log 2>&1
  python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) > 20]
print({r["Name"].split("(")[0].replace("void rm::", "").replace("rm::", "")[:22]: round(float(r["AverageNs"]) / 1e3, 1) for r in rows})
PY
  grep -o '"ms_per_tick": [0-9.]*' $O/run.log
done
rm -rf $R/gpurun_out/prof_knob
