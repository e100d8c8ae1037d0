"""profiles/fetch_calibration.json from a rocprofv3 --pmc FETCH_SIZE pass over tools/fetch_calib: per access pattern the
factor true bytes / reported bytes (see tools/fetch_calib.hip).   python tools/fetch_calib.py <rocprof_dir> <out_json>"""
import collections
import csv
import glob
import json
import os
import sys

acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "calib_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("calib_")[1].split("(")[0]].append(float(r["Counter_Value"]))
true_kib = float(1 << 20)
out = {"commit": os.environ.get("RM_COMMIT", "unrecorded"), "bytes_read_per_kernel": 1 << 30,
       "how": "tools/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE; factor = true bytes / (FETCH_SIZE * 1024)", "patterns": {}}
for k, v in sorted(acc.items()):
    rep = sum(v) / len(v)
    out["patterns"][k] = {"fetch_size_kib_reported": round(rep, 1), "factor": round(true_kib / rep, 3), "dispatches": len(v)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
