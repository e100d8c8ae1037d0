"""Merge the entries of one collection call's pmc_traffic.json into profiles/pmc_traffic.json (an entry = one workload key).
    python tools/pmc_merge.py gpurun_out/r04/pmc_traffic.json [more.json ...]"""
import json
import os
import sys

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
out = json.load(open(dst)) if os.path.exists(dst) else {}
for src in sys.argv[1:]:
    for k, v in json.load(open(src)).items():
        out[k] = v
        print("merged", k, "from", src)
json.dump(out, open(dst, "w"), indent=1)
