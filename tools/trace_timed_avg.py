"""A bench run under `rocprofv3 --kernel-trace`: the kernels' average durations over the launches of the TIMED region only.
rocprofv3's --stats averages every dispatch of a kernel in the process -- set-up and warm-up launches too, which run with
fewer contexts in flight and are shorter -- while the result line's roofline.kernels are the timed launches' (bench.py
samples them with events bound to the dispatches).  This takes the last `steps x contexts-in-rotation` dispatches of
every kernel of the launch sequence from the trace, so that the two can be compared like with like.

    python tools/trace_timed_avg.py <dir with *kernel_trace.csv> <bench line json> <out json>"""
import csv
import glob
import json
import sys


def short(name):
    return name.replace("void ", "").replace("rm::", "").split("(")[0]


def main():
    trace_dir, line_path, out_path = sys.argv[1:4]
    line = json.loads([ln for ln in open(line_path).read().splitlines() if ln.startswith("{")][-1])
    steps = int(line["steps"])
    wanted = {k.split("<")[0] for k in line["roofline"]["kernels"]}
    rows = collections_by_kernel(trace_dir)
    out = {"steps": steps, "note": "average duration [us] of the last `steps` dispatches per launch-sequence kernel in the rocprofv3 kernel "
                                   "trace of the same command (the timed region), beside the line's own event-timed averages",
           "kernels": {}}
    for name, durs in rows.items():
        base = name.split("<")[0]
        if base not in wanted:
            continue
        # the sequential leg and the host-transfer legs launch other kernels; a kernel of the sequence is launched once per step
        timed = durs[-steps:] if len(durs) >= steps else durs
        line_us = line_kernel_us(line, name)
        out["kernels"][name] = {"dispatches_in_trace": len(durs), "timed_avg_us": sum(timed) / len(timed), "all_avg_us": sum(durs) / len(durs),
                                "line_avg_us": line_us, "line_over_trace": (line_us / (sum(timed) / len(timed))) if line_us else None}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


def line_kernel_us(line, trace_name):
    """the line's average for the kernel rocprofv3 calls `trace_name`: the same name (template arguments as the launch site spells
    them may differ from the trace's -- RM_MODEL_LOGDIST / 4), else the only kernel of the same base name"""
    ks = line["roofline"]["kernels"]
    norm = lambda n: n.replace(" ", "")
    for k, v in ks.items():
        if norm(k) == norm(trace_name):
            return v["avg_us"]
    same = [v["avg_us"] for k, v in ks.items() if k.split("<")[0] == trace_name.split("<")[0]]
    return same[0] if len(same) == 1 else None


def collections_by_kernel(trace_dir):
    rows = {}
    for f in glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True):
        recs = []
        for r in csv.DictReader(open(f)):
            recs.append((int(r["Start_Timestamp"]), short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        recs.sort()
        for _, k, d in recs:
            rows.setdefault(k, []).append(d)
    return rows


if __name__ == "__main__":
    main()
