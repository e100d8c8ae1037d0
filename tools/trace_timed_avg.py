"""A bench run under `rocprofv3 --kernel-trace`: the kernels' average durations over the launches of the TIMED region only.
rocprofv3's --stats averages every dispatch of a kernel in the process -- set-up and warm-up launches too, which run with
fewer contexts in flight and are shorter -- while the result line carries two sets of intervals (events bound to the
dispatches): roofline.kernels from launch sequences that ran ALONE after the timed region, roofline.contended.kernels from the
timed region.  This takes the same dispatches of every kernel of the launch sequence from the trace -- the last 2k (the alone
pass) and the `steps` before them -- so that like is compared with like.  (For commands without the extra legs: --no-host-transfer
--no-scale-probe, as tools/collect_profiles.sh runs them; the legs launch the same kernels again.)

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
    rl = line["roofline"]
    wanted = {k.split("<")[0] for k in rl["kernels"]}
    alone = rl.get("alone") or {}
    k_alone = int(alone.get("sequences") or 0)   # the alone pass: k unprobed sequences, then k probed ones, after the timed region
    rows = collections_by_kernel(trace_dir)
    out = {"steps": steps, "alone_sequences": k_alone,
           "note": "average durations [us] per launch-sequence kernel in the rocprofv3 kernel trace of the same command: the dispatches of "
                   "the ALONE pass (the last 2k: k unprobed, then k probed -- roofline.kernels are the probed ones' own intervals) and of "
                   "the TIMED region before them (roofline.contended.kernels), beside the line's own event-timed averages",
           "kernels": {}}
    for name, durs in rows.items():
        base = name.split("<")[0]
        if base not in wanted:
            continue
        tail = 2 * k_alone
        a_plain = durs[len(durs) - tail:len(durs) - k_alone] if tail and len(durs) >= tail else []
        a_probed = durs[len(durs) - k_alone:] if tail and len(durs) >= tail else []
        timed = durs[max(0, len(durs) - tail - steps):len(durs) - tail] if len(durs) > tail else durs
        avg = lambda v: (sum(v) / len(v)) if v else None
        line_alone = line_kernel_us(rl["kernels"], name)
        line_cont = line_kernel_us((rl.get("contended") or {}).get("kernels") or {}, name) if tail else line_alone
        ent = {"dispatches_in_trace": len(durs), "trace_alone_probed_avg_us": avg(a_probed), "trace_alone_unprobed_avg_us": avg(a_plain),
               "trace_timed_avg_us": avg(timed), "all_avg_us": avg(durs), "line_alone_avg_us": line_alone, "line_contended_avg_us": line_cont}
        if line_alone and avg(a_probed):
            ent["line_over_trace_alone"] = line_alone / avg(a_probed)
        if line_cont and avg(timed):
            ent["line_over_trace_timed"] = line_cont / avg(timed)
        out["kernels"][name] = ent
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


def line_kernel_us(ks, trace_name):
    """the line's average for the kernel rocprofv3 calls `trace_name`: the same name (template arguments as the launch site spells
    them may differ from the trace's -- RM_MODEL_LOGDIST / 4), else the only kernel of the same base name"""
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
