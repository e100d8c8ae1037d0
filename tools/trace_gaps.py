"""Timeline summary of a rocprofv3 --kernel-trace CSV: busy time, span, and the largest idle gaps."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[len(rows) // 3:]          # skip warm-up
span = rows[-1][1] - rows[0][0]
# union of busy intervals
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, name, q in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels %d span %.1f us busy %.1f us (%.0f%%)" % (len(rows), span / 1e3, busy / 1e3, 100.0 * busy / span))
gaps.sort(reverse=True)
print("largest gaps (us, next kernel):", [(round(g / 1e3, 1), n) for g, n in gaps[:12]])
import collections
by = collections.Counter()
for g, n in gaps:
    by[n] += g
print("gap time before kernel (us):", {k: round(v / 1e3, 1) for k, v in by.most_common(8)})
