"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter group per
pass -- TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2, MI355X_MICROARCH.md "rocprofv3 PMC slots").

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <workload> <ticks_per_launch> <out_csv> <out_json> [<sq_dir> [<tracked_csv>]]

<tracked_csv>: where the CSV will live in the repository (profiles/...): the path the JSON entries cite.

RM_COMMIT=<id> stamps the commit the counters were collected for into the JSON (the GPU box has no .git).

<sq_dir>: a third pass with --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY: the stages' vector
issue time.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs (MI355X_MICROARCH.md), so
issue time on the whole chip = 4 * count / 1024 SIMDs / 2.4 GHz (tools/clockprobe.hip: 2.4 GHz held).

bytes = (factor * FETCH_SIZE + WRITE_SIZE) * 1024, both counters in KiB.  On gfx950 FETCH_SIZE counts the 128-byte
requests of a wide coalesced stream at 64 bytes (MI355X_MICROARCH.md section HBM: double it) -- and "other access widths
are uncalibrated", so the factor of every stage comes from profiles/fetch_calibration.json (tools/fetch_calib.hip under
the same counter): 2.0 for the streaming stages (k_filter: 16-byte pre-filter records; k_reorder: runs of a frame's
records; k_self_entries), 1.0 for the gather stages (k_exact, k_sinr: aligned 32- and 64-byte records at scattered
places move one 64-byte request each, which is what the counter reports).  The one-launch tick (k_tick_frames) streams
the near groups' records and gathers the candidates' in one kernel: both readings are written (min = factor 1, the
headline figure = factor 2, an upper bound).  Kernels are grouped into the stages bench.py brackets with HIP events; the
JSON is what bench.py reads for roofline.traffic.
"""
import collections
import csv
import glob
import json
import os
import sys

STAGE = (("k_dense_count<0>", "k_dense_count_null"), ("k_dense_write<0>", "k_dense_write_null"), ("k_dense_count<1>", "k_dense_count_udgm"),
         ("k_dense_write<1>", "k_dense_write_udgm"), ("k_dense_write", "k_dense_write"), ("k_dense_scan", "k_dense_scan"), ("k_ov_pairs", "k_ov_pairs"), ("k_ov_exact", "k_ov_exact"), ("k_ov_verdict", "k_ov_verdict"), ("k_ov_", "k_ov_index"),
         ("k_sinr_scan", "k_sinr_scan"), ("k_tick_frames", "k_tick_frames"), ("k_frames_cand", "k_filter"), ("k_tick_prep", "k_filter"), ("k_filter", "k_filter"), ("k_near_pairs", "k_filter"), ("k_exact", "k_exact"),
         ("k_reorder", "k_reorder"), ("k_self_entries", "k_self_entries"), ("k_sinr", "k_sinr"),
         ("k_cell_off", "k_cell_off+k_slot_scan"), ("k_slot_scan", "k_cell_off+k_slot_scan"), ("k_finalize", "k_finalize"))


# access pattern of a stage's reads -> calibration entry (profiles/fetch_calibration.json)
PATTERN = {"k_dense_count_null": "stream4", "k_dense_write_null": "stream4", "k_dense_count_udgm": "stream16", "k_dense_write_udgm": "stream16", "k_dense_write": "stream16",
           "k_dense_scan": "stream4", "k_ov_pairs": "gather32", "k_ov_exact": "gather32", "k_ov_verdict": "stream16", "k_ov_index": "stream16", "k_sinr_scan": "gather32",
           "k_filter": "stream16", "k_reorder": "runs8", "k_self_entries": "stream4", "k_exact": "gather32", "k_sinr": "gather32",
           "k_finalize": "gather32", "k_cell_off+k_slot_scan": "stream4", "k_tick_frames": "stream16"}


def fetch_factor(stage):
    try:
        cal = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "fetch_calibration.json")))
        pat = PATTERN.get(stage, "stream16")
        return max(1.0, float(cal["patterns"][pat]["factor"])), pat, cal.get("commit", "unrecorded")
    except (OSError, ValueError, KeyError):
        return 2.0, "uncalibrated (the guide's factor for wide streams)", "none"


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
    tracked_csv = sys.argv[8] if len(sys.argv) > 8 else os.path.relpath(out_csv)
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    valu = read(sq_dir, "SQ_ACTIVE_INST_VALU") if sq_dir else {}
    insts = read(sq_dir, "SQ_INSTS_VALU") if sq_dir else {}
    thr_cyc = read(sq_dir, "SQ_THREAD_CYCLES_VALU") if sq_dir else {}
    inst_cyc = read(sq_dir, "SQ_INST_CYCLES_VALU") if sq_dir else {}
    valu_stage = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0.0])
    for k in valu:
        if len(valu[k]) > 8:
            for prefix, stage in STAGE:
                if k.startswith(prefix):
                    valu_stage[stage][0] += sum(valu[k]) / len(valu[k])
                    valu_stage[stage][1] += sum(insts.get(k, [0])) / max(1, len(insts.get(k, [0])))
                    valu_stage[stage][2] += sum(thr_cyc.get(k, [0])) / max(1, len(thr_cyc.get(k, [0])))
                    valu_stage[stage][3] += sum(inst_cyc.get(k, [0])) / max(1, len(inst_cyc.get(k, [0])))
                    break
    rows, stages = [], collections.defaultdict(lambda: [0.0, 0.0])
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [0])))
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
        n = max(len(fetch.get(k, [])), len(write.get(k, [])))
        fac = 2.0
        for prefix, stage in STAGE:
            if k.startswith(prefix):
                fac = fetch_factor(stage)[0]
                break
        rows.append((k, n, f, w, fac, int((fac * f + w) * 1024)))
        if n > 8:   # per-tick kernels only (set-up kernels run once)
            for prefix, stage in STAGE:
                if k.startswith(prefix):
                    stages[stage][0] += f
                    stages[stage][1] += w
                    break
    with open(out_csv, "w") as fh:
        fh.write("kernel,dispatches,FETCH_SIZE_KB_avg_raw,WRITE_SIZE_KB_avg,fetch_factor,hbm_bytes_per_launch_corrected\n")
        for r in rows:
            fh.write("%s,%d,%.1f,%.1f,%.2f,%d\n" % r)
    try:
        out = json.load(open(out_json))
    except (OSError, ValueError):
        out = {}
    src = ("%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --inflight 1, %s ticks per "
           "launch), (factor*FETCH_SIZE + WRITE_SIZE)*1024, factor per access pattern from profiles/fetch_calibration.json"
           % (tracked_csv, tpl))
    out[workload] = {"ticks_per_launch": int(tpl), "commit": os.environ.get("RM_COMMIT", "unrecorded"), "csv": tracked_csv}
    for stage, (f, w) in stages.items():
        if stage in ("k_tick_frames", "k_sinr_scan") and int(tpl) != 1:
            continue   # the lone tick's kernels belong to the ticks_per_launch = 1 passes (the batch runs contain bench.py's sequential leg)
        if stage not in ("k_tick_frames", "k_sinr_scan") and int(tpl) == 1 and ("k_tick_frames" in stages) and workload.endswith("_tick"):
            continue   # ... and the sweep's kernels (set-up launches of a tick run) do not belong to them
        fac, pat, cal_commit = fetch_factor(stage)
        out[workload][stage] = {"hbm_bytes_per_launch": int((fac * f + w) * 1024), "fetch_size_kb_raw": round(f, 1),
                                "write_size_kb": round(w, 1), "fetch_factor": fac, "fetch_pattern": pat,
                                "calibration_commit": cal_commit, "source": src}
        if stage == "k_tick_frames":   # streams and gathers in one kernel: the other reading as well
            out[workload][stage]["hbm_bytes_per_launch_min"] = int((f + w) * 1024)
    for stage, (quad, n_inst, thr, icyc) in valu_stage.items():
        if stage in out[workload]:
            out[workload][stage]["valu_issue_us_per_launch"] = round(4.0 * quad / 1024.0 / 2400.0, 2)
            out[workload][stage]["valu_instructions_per_launch"] = int(n_inst)
            if thr > 0 and icyc > 0:   # active lanes per issued vector instruction cycle, of 64
                out[workload][stage]["valu_lane_utilisation"] = round(thr / (64.0 * icyc), 3)
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out[workload], indent=1))


if __name__ == "__main__":
    main()
