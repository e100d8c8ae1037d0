"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter group per
pass -- TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2, MI355X_MICROARCH.md "rocprofv3 PMC slots").

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <workload> <ticks_per_launch> <out_csv> <out_json> [<sq_dir>]

RM_COMMIT=<id> stamps the commit the counters were collected for into the JSON (the GPU box has no .git).

<sq_dir>: a third pass with --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY: the stages' vector
issue time.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs (MI355X_MICROARCH.md), so
issue time on the whole chip = 4 * count / 1024 SIMDs / 2.4 GHz (tools/clockprobe.hip: 2.4 GHz held).

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes
(MI355X_MICROARCH.md section HBM), both counters are in KiB.  Kernels are grouped into the stages
bench.py brackets with HIP events; the JSON is what bench.py reads for roofline.traffic.
"""
import collections
import csv
import glob
import json
import os
import sys

STAGE = (("k_tick_frames", "k_tick_frames"), ("k_frames_cand", "k_filter"), ("k_tick_prep", "k_filter"), ("k_filter", "k_filter"), ("k_near_pairs", "k_filter"), ("k_exact", "k_exact"),
         ("k_reorder", "k_reorder"), ("k_self_entries", "k_self_entries"), ("k_sinr", "k_sinr"),
         ("k_cell_off", "k_cell_off+k_slot_scan"), ("k_slot_scan", "k_cell_off+k_slot_scan"), ("k_finalize", "k_finalize"))


def short(name):
    n = name.replace("void ", "").replace("rm::", "")
    return n.split("(")[0]


def read(path, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch_dir, write_dir, workload, tpl, out_csv, out_json = sys.argv[1:7]
    sq_dir = sys.argv[7] if len(sys.argv) > 7 else None
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    valu = read(sq_dir, "SQ_ACTIVE_INST_VALU") if sq_dir else {}
    insts = read(sq_dir, "SQ_INSTS_VALU") if sq_dir else {}
    valu_stage = collections.defaultdict(lambda: [0.0, 0.0])
    for k in valu:
        if len(valu[k]) > 8:
            for prefix, stage in STAGE:
                if k.startswith(prefix):
                    valu_stage[stage][0] += sum(valu[k]) / len(valu[k])
                    valu_stage[stage][1] += sum(insts.get(k, [0])) / max(1, len(insts.get(k, [0])))
                    break
    rows, stages = [], collections.defaultdict(lambda: [0.0, 0.0])
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [0])))
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
        n = max(len(fetch.get(k, [])), len(write.get(k, [])))
        rows.append((k, n, f, w, int((2 * f + w) * 1024)))
        if n > 8:   # per-tick kernels only (set-up kernels run once)
            for prefix, stage in STAGE:
                if k.startswith(prefix):
                    stages[stage][0] += f
                    stages[stage][1] += w
                    break
    with open(out_csv, "w") as fh:
        fh.write("kernel,dispatches,FETCH_SIZE_KB_avg_raw,WRITE_SIZE_KB_avg,hbm_bytes_per_launch_corrected\n")
        for r in rows:
            fh.write("%s,%d,%.1f,%.1f,%d\n" % r)
    try:
        out = json.load(open(out_json))
    except (OSError, ValueError):
        out = {}
    src = ("%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --inflight 1, %s ticks per "
           "launch), (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE reads half of a wide "
           "coalesced stream)" % (os.path.relpath(out_csv), tpl))
    out[workload] = {"ticks_per_launch": int(tpl), "commit": os.environ.get("RM_COMMIT", "unrecorded")}
    for stage, (f, w) in stages.items():
        if (stage == "k_tick_frames") != (int(tpl) == 1):
            continue   # the one-launch tick belongs to the ticks_per_launch = 1 passes (the batch runs contain bench.py's sequential leg)
        out[workload][stage] = {"hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_size_kb_raw": round(f, 1),
                                "write_size_kb": round(w, 1), "source": src}
    for stage, (quad, n_inst) in valu_stage.items():
        if stage in out[workload]:
            out[workload][stage]["valu_issue_us_per_launch"] = round(4.0 * quad / 1024.0 / 2400.0, 2)
            out[workload][stage]["valu_instructions_per_launch"] = int(n_inst)
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out[workload], indent=1))


if __name__ == "__main__":
    main()
