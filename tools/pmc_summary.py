"""Summarise a rocprofv3 --pmc counter_collection.csv: mean of every counter per kernel name."""
import csv, sys, glob, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not k.startswith(("void rm::", "rm::")):
        continue
    n = max(len(v) for v in cs.values())
    print("%s  (%d dispatches)" % (k[:90], n))
    for c, v in sorted(cs.items()):
        print("    %-24s %14.0f" % (c, sum(v) / len(v)))
