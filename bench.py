#!/usr/bin/env python3
"""bench.py -- Tx->Rx link evaluations per second of the MI355X radio-medium engine.

A "step" is one pass of the hot path over one batch of synthetic input: ONE launch sequence that
sweeps `ticks_per_step` simulated ticks (128 on one GPU, 64 per rank and at most 512 on several; `--batch`).  In
every tick T = 1% of N nodes transmit a 127-byte frame; the engine evaluates all T x (N-1) links
(log-distance path loss + log-normal shadowing, BASELINE.json configs[2]: 100k nodes, 1% concurrent
Tx) and leaves the ordered heard-link records (receiver, rssi, verdict) in HBM, one result slot per
tick.  Inputs (node state, every tick's source list) are resident in HBM before the timed region
starts.  `value` = link evaluations of the K timed steps / their wall time, whatever K is.

The benchmarked medium carries no state from tick to tick (no on-air list, no random draws with
the reference's default probabilities; RadioMedium.transmit treats every packet on its own), and a
lone tick of this size is a 13 us launch on a mostly idle device.
So `--batch` ticks (default 128) go through ONE launch sequence (rm_batch_run_sources_device), and
`--inflight` contexts (default 3), each with its own stream, take the batches in turn.  The
strictly sequential rate (one tick at a time, what a closed-loop simulation sees) is measured in
the same run and printed as "sequential_ticks".

    python bench.py [--gpus N --steps K --warmup W] [--workload c2|c3|udgm|m1|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: receivers are partitioned over the ranks (regions of the plane); per batch each rank packs the source indices of
the transmitters it owns for all ticks of the batch, the ranks all-gather them over RCCL/xGMI (ONE
collective per batch, on the context's own stream and communicator; the other contexts' sweeps run
under it) and every rank sweeps the gathered frames against its receivers.  Default scaling is
STRONG: the BASELINE config itself (100k nodes for configs[2]; `--workload c4` for the 8-GPU config),
its receivers split over the ranks.  `--scaling weak` grows the node count as 100k x sqrt(N) at constant
density and Tx fraction instead, so that the link evaluations per GPU and tick stay those of the 1-GPU
config; `--as-rank R:W` runs one rank's share of a W-GPU run (strong scaling, or weak with --scaling weak) on one GPU, without the
collective.  Rank 0 prints ONE JSON line.

Map of this file.  What the driver runs is `python bench.py` with no flags (and `--gpus N --steps K --warmup W`): main() ->
measure(): workload c3 = BASELINE configs[2], 128 ticks per launch, three contexts; its timed region is the K steps between
fence() and fence() (barrier + synchronize on both sides, MAX over ranks), and `value`, `ms_per_step`, `roofline` come from it
alone.  Everything else on the line is an extra key measured AFTER the timed region, each by its own function, and none of it
enters `value`:
    roofline_object()     the `roofline` object of a run (kernel intervals from the library's probes, 8(d) bytes, cross-check)
    cpu_baseline()        `cpu_baseline`: the oracle (oracle/rm_oracle.c) on a bounded sample, rank 0, one GPU only
    sequential (inline)   `sequential_ticks`: the same ticks one at a time on one context
    host_transfer_legs()  `with_host_transfer`: the PCIe-inclusive closed loops (never `value`)
    scale_probe()         `at_1M_nodes`: a short run at a million nodes
    dense_probe()         `dense_layout`: the reference's default Null medium and a unit disc over the whole field
Modes behind flags (none of them the driver's path): --workload (the other BASELINE configs), --as-rank R:W (one rank's share on
one GPU), --dense-only, --force-sharded / --collective / --partition (the multi-GPU driver on one GPU, rehearsals), --nodes,
--link-capacity, --batch, --inflight.  spawn_ranks() starts the ranks of `--gpus N` when no launcher did; dry_run() is the
CPU-only rehearsal of the launch contract (tests/test_bench_launch.py).
"""
import argparse
import copy
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
S_NODE, S_TX, S_REC = 37, 56, 25   # algorithmic bytes: SURVEY.md section 8(d)

WORKLOADS = {
    # name: (config index, N, tx fraction, model name, description)
    "c2": (2, 10_000, 0.01, "logdist", "10k nodes, 1% concurrent-Tx, log-distance path loss"),
    "c3": (3, 100_000, 0.01, "logdist_shadow", "100k nodes, 1% concurrent-Tx, log-distance + log-normal shadowing"),
    "udgm": (3, 100_000, 0.01, "udgm", "100k nodes, 1% concurrent-Tx, reference UDGM (unit disc)"),
    "udgm_lossy": (3, 100_000, 0.01, "udgm_lossy", "100k nodes, 1% concurrent-Tx, reference UDGM with successRatioRx = 0.9 "
                                                   "(every heard link consumes a java.util.Random draw)"),
    "c3x6": (3, 100_000, 0.06, "logdist_shadow", "100k nodes, 6% concurrent-Tx (six ticks' worth of frames in one pass)"),
    "m1": (5, 1_000_000, 0.001, "logdist_shadow", "1M nodes, 0.1% concurrent-Tx, log-distance + log-normal shadowing"),
    "m1x": (5, 1_000_000, 0.01, "logdist_shadow", "1M nodes, 1% concurrent-Tx, log-distance + log-normal shadowing"),
    # the 8-GPU configs of BASELINE.json, runnable on one GPU as well (parity cases, not the headline):
    "c4": (4, 100_000, 0.05, "logdist_sinr16", "100k nodes, 5% concurrent-Tx, 16 channels with co-channel SINR capture"),
    "c5": (5, 1_000_000, 0.001, "logdist_sinr_overlap", "1M nodes, 0.1% new Tx per tick, multi-tick packet overlap (SINR)"),
}
# per-workload overrides: 16 channels; tick length (c4: one frame time, so the 5% are the concurrent set)
# (c5: a result slot per tick of a batch holds ~45 k heard links; the per-receiver lists of RM_SINR_SCAN=0 want 2^25 entries: --link-capacity)
# (c4: a tick of 5000 frames has ~150 k candidates at the interference floor and ~14 k heard links; the interference is summed per
# receiver -- no list entries -- so a result slot of 2^20 holds it and 128 ticks go through one launch sequence.  RM_SINR_ACC=0 -- the
# per-receiver lists -- wants --link-capacity 8388608 --batch 32)
EXTRA = {"c3x6": dict(link_capacity=1 << 22), "c4": dict(channels16=True, tick_us=8128, link_capacity=1 << 20, batch=128),
         "c5": dict(link_capacity=1 << 18, batch=128)}


def baseline_metric():
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, ValueError, KeyError):
        return "Tx->Rx link evaluations/sec at N nodes, 1% concurrent-Tx; 1/2/4/8 GPUs"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (launch sequences of --batch ticks); default: 1920 ticks' worth")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed steps before them; default: 192 ticks' worth")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense-only", action="store_true", help="only the dense layouts (nothing to cull: unit disc over the whole field, "
                                                              "the reference's default Null medium), as the result line")
    ap.add_argument("--cpu-sample-ticks", type=float, default=5.0)
    ap.add_argument("--inflight", type=int, default=0,
                    help="engine contexts per GPU, each with its own stream, taking the batches in turn (ticks are "
                         "independent for the media without an on-air list); default 3 (1 for the media with frames that stay on the air)")
    ap.add_argument("--batch", type=int, default=0,
                    help="ticks per launch sequence (rm_batch_run_sources_device, at most 512); 1 = one tick per sequence; "
                         "default 128 (the workload's own for c4 / c5), 64 per rank up to 512 with several GPUs (one all-gather per "
                         "batch: fewer, larger collectives, and the stream hand-over around a collective is amortised over more ticks)")
    ap.add_argument("--as-rank", default="", metavar="R:W",
                    help="one process, no collective: sweep the weak-scaling workload of W ranks against the receiver "
                         "range of rank R only (what one GPU of a W-GPU run computes per tick)")
    ap.add_argument("--link-capacity", type=int, default=0, help="override the workload's link capacity per result slot")
    ap.add_argument("--nodes", type=int, default=0, help="override the workload's node count (same density and Tx fraction)")
    ap.add_argument("--no-host-transfer", action="store_true", help="skip the PCIe-inclusive legs (with_host_transfer)")
    ap.add_argument("--no-scale-probe", action="store_true",
                    help="skip the short 1M-node run that shows the sweep's HBM fraction at scale")
    ap.add_argument("--no-weak-probe", action="store_true", help="several GPUs: skip the extra weak-scaling pass (key weak_scaling)")
    ap.add_argument("--spatial-ids", action="store_true",
                    help="experiment: number the synthetic nodes along a space-filling curve, so that a rank's range of node "
                         "indices is a region of the area (NOT the BASELINE layout's numbering; the link count is the same)")
    ap.add_argument("--partition", default="spatial", choices=["spatial", "index"],
                    help="several GPUs / --as-rank: a rank's receivers are a REGION of the plane (the k-d split of all positions: "
                         "its filter drops the frames far from the region; default) or a range of node indices")
    ap.add_argument("--collective", default="auto", choices=["auto", "lib", "torch"],
                    help="several GPUs: who runs the all-gather of Tx records -- lib: libradiomedium_hip.so itself (RCCL bound inside "
                         "the library: pack + ncclAllGather + sweep are ONE call per batch), torch: torch.distributed around the "
                         "engine calls (the gloo rehearsal on one GPU needs it); auto: lib with the nccl backend")
    ap.add_argument("--host-cull", type=float, default=0.0, metavar="METRES",
                    help="experiment (--as-rank): hand the rank only the frames within METRES of its region's bounding box, as if "
                         "the frame list had been compacted already -- what rank-level frame culling can buy at most")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="several GPUs: strong (default) = the BASELINE config itself, receivers split over the ranks; "
                         "weak = node count grown as sqrt(GPUs) so that the link evaluations per GPU stay fixed")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the pipelined multi-GPU tick driver even on one GPU (testing)")
    ap.add_argument("--profile-every", type=int, default=16,
                    help="HIP-event sample of the dominant kernel every n-th tick of the timed region")
    return ap.parse_args()


# a kernel (base name as rocprofv3 prints it) -> the stage whose algorithmic bytes price it, and its key in
# profiles/pmc_traffic.json (tools/pmc_traffic.py groups the counters the same way)
KERNEL_STAGE = (("k_dense_write", "dense"), ("k_dense_count", "dense_count"), ("k_tick_frames_scan", "tick"), ("k_tick_frames", "tick"), ("k_sinr_scan", "sinr_scan"), ("k_ov_pairs", "ov_pairs"),
                ("k_ov_exact", "ov_exact"), ("k_ov_verdict", "ov_verdict"), ("k_filter", "filter"),
                ("k_frames_cand", "filter"), ("k_exact", "exact"), ("k_reorder", "reorder"), ("k_sinr", "sinr"),
                ("k_self_entries", "sinr"))
# (the interference stages of a batch of overlapping SINR ticks, rm_airbatch.hip: ov_pairs reads per heard link its receiver's
# records and per new frame ~its near frames' index entries, and writes the surviving pairs; ov_exact gathers a 32-byte receiver
# record and a 64-byte frame record per pair and adds to the link's 16-byte sum; ov_verdict reads sum, rssi, flag and writes sinr, verdict)
PMC_KEY = {"filter": "k_filter", "exact": "k_exact", "reorder": "k_reorder", "sinr": "k_sinr", "tick": "k_tick_frames",
           "sinr_scan": "k_sinr_scan"}


def kernel_stage(name):
    for prefix, stage in KERNEL_STAGE:
        if name.startswith(prefix):
            return stage
    return None


def required_bytes(n_loc, t_per_tick, heard, ticks_per_launch, sinr_column, tile_reuse=1):
    """What ONE launch sequence of this shape has to move at the least: section 8(d)'s bytes with the two terms the launch shape
    changes -- a heard-link record is 17 bytes where the medium has no sinr column (rm_device_result.sinr is NULL: the column
    is neither allocated nor written), and a filter workgroup that keeps its receivers for `tile_reuse` ticks of the launch
    (rm_filter.hip: filter_wg_body) fetches the receiver table once per that many ticks, not once per tick."""
    rec = S_REC if sinr_column else S_REC - 8
    return int((n_loc * S_NODE / max(1, tile_reuse) + t_per_tick * S_TX + heard * rec) * ticks_per_launch)


def tile_reuse_of(n_loc, ticks_per_launch):
    """ticks a filter workgroup sweeps with one load of its 1024 receivers (launch_filter_batch's rule, rm_filter.hip)"""
    tiles = -(-n_loc // 1024)
    return max(1, min(ticks_per_launch, (tiles * ticks_per_launch) // 3072)) if ticks_per_launch > 1 else 1


def kernel_table(kernels, n_samples, own):
    """per kernel: average interval, launches per sequence, its own algorithmic bytes"""
    out = {}
    for name, k in kernels.items():
        if k["launches"] == 0:
            continue
        us = k["ms"] / k["launches"] * 1e3
        st = kernel_stage(name)
        b = own.get(st)
        out[name] = {"avg_us": us, "launches_sampled": k["launches"], "launches_per_sequence": k["launches"] / max(1, n_samples), "stage": st,
                     "algorithmic_bytes": b, "achieved_GBps": (b / (us * 1e-6) / 1e9) if (b and us > 0) else None,
                     "hbm_frac": (b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (b and us > 0) else None}
    return out


def roofline_object(kernels, n_samples, n_loc, t_per_tick, heard, cand, ticks_per_launch, step_s, contexts, workload, pmc_ok=True,
                    tick_key=False, pairs_per_launch=0, under_profiler=None, alone=None, sinr_column=False, tile_reuse=1, required=None):
    """The result line's `roofline`, for the DOMINANT kernel of a launch sequence:

      achieved = bytes of ONE launch sequence / the dominant kernel's average duration, frac = achieved / 8 TB/s.

    The duration is the kernel's OWN dispatch interval (a pair of HIP events bound to that dispatch: rm_profile_enable ->
    hipExtLaunchKernelGGL -- what `rocprofv3 --kernel-trace --stats` reports for the same kernel), taken from launch sequences
    that run ALONE on the device (`alone`: probed sequences on one context after the timed region, nothing else in flight), so
    an interval is the kernel's time, not the time it shared with other contexts' kernels.  The bytes: `required_bytes` --
    section 8(d)'s figure for this launch shape (required_bytes(): 17-byte records without a sinr column, a receiver tile
    charged once per reuse window); the plain 8(d) figure is carried beside it (`algorithmic_bytes_per_launch`, `frac_8d`).
    No kernel can move its launch's required bytes faster than the HBM: frac > 1 refuses the line.

    `bound` names the resource the STEP uses most of: `hbm` (bytes of the step / driver-timed step / 8 TB/s) or `valu` (the
    kernels' vector-issue time, SQ_ACTIVE_INST_VALU from the PMC passes on file, / the driver-timed step) -- SURVEY.md 8(d):
    these culled sweeps are issue-bound, not HBM-bound, and the line says so.
    `contended`: the same kernels' intervals as sampled INSIDE the timed region (several contexts in flight: every interval is
    stretched by the others' kernels).  Kept only for the cross-check: their sum over the contexts in flight cannot exceed the
    driver-timed step by more than the probes' own cost, which is measured in the same run (`alone.probe_cost_us_per_launch`:
    the same sequences with and without probes)."""
    b_tick = n_loc * S_NODE + t_per_tick * S_TX + heard * S_REC
    b_launch = b_tick * ticks_per_launch
    b_req = required_bytes(n_loc, t_per_tick, heard, ticks_per_launch, sinr_column, tile_reuse) if required is None else int(required[0])
    rec_w = 17 if not sinr_column else 25
    own = {"filter": (n_loc * 16 // max(1, tile_reuse) + t_per_tick * 28 + cand * 12) * ticks_per_launch,
           "exact": (cand * 44 + heard * 13) * ticks_per_launch,
           "reorder": heard * (13 + rec_w) * ticks_per_launch,
           "tick": b_req,
           "ov_pairs": (heard * 49 + t_per_tick * 64 * 48) * ticks_per_launch + pairs_per_launch * 16,
           "ov_exact": pairs_per_launch * (16 + 32 + 64 + 16),
           "ov_verdict": heard * 34 * ticks_per_launch,
           # the dense tick (rm_dense.hip): the count pass reads the node-ordered columns, the write pass writes the records
           "dense": heard * rec_w * ticks_per_launch, "dense_count": n_loc * S_NODE + t_per_tick * S_TX}
    cont = kernel_table(kernels, n_samples, own)
    src = alone if (alone and alone.get("kernels")) else None
    per_kernel = kernel_table(src["kernels"], src["n_samples"], own) if src else cont
    priced = {n: v for n, v in per_kernel.items() if v["stage"] is not None}
    if priced:
        dominant = max(priced, key=lambda n: priced[n]["avg_us"] * priced[n]["launches_per_sequence"])
        kern_us = priced[dominant]["avg_us"]
    else:
        dominant, kern_us = None, 0.0
    achieved = b_req / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
    frac = achieved / HBM_PEAK_GBS
    frac_8d = (b_launch / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if kern_us > 0 else None
    if frac > 1.0:
        raise SystemExit("roofline refused: %s would have moved the launch's required bytes (%d: %d receivers, %d frames and %d heard links "
                         "per tick, %d ticks) in %.1f us = %.2f x the HBM peak"
                         % (dominant, b_req, n_loc, t_per_tick, heard, ticks_per_launch, kern_us, frac))
    seq_us = sum(v["avg_us"] * v["launches_per_sequence"] for v in cont.values())
    launches_per_seq = sum(v["launches_per_sequence"] for v in cont.values())
    share_us = seq_us / max(1, contexts)
    probe_cost = (alone or {}).get("probe_cost_us_per_launch")
    probe_slack_us = (probe_cost if probe_cost is not None else 0.0) * launches_per_seq / max(1, contexts)
    if under_profiler is None:   # (a profiler stretches every launch: the bench's own cross-check is for runs without one)
        under_profiler = any("rocprof" in os.environ.get(v, "") for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH"))
    check_ok = bool(share_us <= step_s * 1e6 * 1.02 + probe_slack_us) if cont else None
    hbm_step = b_req / step_s / 1e9 / HBM_PEAK_GBS if step_s > 0 else 0.0
    rl = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
          "traffic": None, "kernel": dominant, "kernel_avg_us": kern_us,
          "kernel_time_source": ("probed launch sequences alone on the device, after the timed region (one context, nothing else in flight)"
                                 if src else "probed launch sequences of the timed region (contended: no separate pass in this mode)") +
                                ": HIP events bound to the kernel's own dispatch (hipExtLaunchKernelGGL start/stop), the interval "
                                "rocprofv3 --kernel-trace reports",
          "launches_sampled": (src["n_samples"] if src else n_samples),
          "required_bytes_per_launch": b_req,
          "required_bytes": required[1] if required is not None else
                            "(N_loc*37/%d + T*56 + H*%d) x %d ticks: 8(d) for this launch shape (%s; a receiver tile is fetched once per %d "
                            "ticks of the launch)" % (max(1, tile_reuse), rec_w, ticks_per_launch,
                                                      "25-byte records" if sinr_column else "17-byte records: no sinr column without the SINR extension",
                                                      max(1, tile_reuse)),
          "algorithmic_bytes_per_launch": b_launch,
          "algorithmic_bytes_per_tick": "N_loc*37 + T*56 + H*25 = %d" % b_tick,
          # (the plain 8(d) bytes over the same interval; above 1 the figure charges bytes this launch shape never moves -- tiles
          # fetched once per reuse window, columns it does not write -- and is left out rather than printed as a fraction of a peak)
          "frac_8d": frac_8d if (frac_8d is not None and frac_8d <= 1.0) else None,
          "kernels": per_kernel,
          "step": {"hbm_frac": hbm_step, "valu_frac": None,
                   "what": "the driver-timed step's use of the two resources: required bytes / step / 8 TB/s, and the kernels' vector-issue "
                           "time / step; `bound` names the larger"},
          "contended": {"kernels": cont, "sequence_kernel_us": seq_us, "contexts_in_flight": contexts, "launches_sampled": n_samples} if src else None,
          "overlap_check": {"kernel_us_per_sequence": seq_us, "contexts_in_flight": contexts, "per_context_share_us": share_us,
                            "step_us": step_s * 1e6, "ok": check_ok, "probe_cost_us_per_launch": probe_cost,
                            "probe_slack_us": probe_slack_us, "under_profiler": bool(under_profiler),
                            "what": "the kernel intervals of one launch sequence of the timed region, summed, over the contexts in flight: "
                                    "at most the driver-timed step x 1.02 + what the probes cost the sampled launches (measured in this "
                                    "run: the same sequences with and without probes)"},
          "whole_step": {"algorithmic_bytes": b_launch, "required_bytes": b_req, "ms_per_step": step_s * 1e3,
                         "achieved": b_req / step_s / 1e9, "frac": hbm_step,
                         "note": "required bytes of one step over the driver-timed step (kernels + launch gaps + host)"},
          "valu_issue": None, "traffic_source": "no PMC pass on file for this workload and launch shape"}
    if alone:
        rl["alone"] = {k: v for k, v in alone.items() if k != "kernels"}
        au = alone.get("step_us_probed")
        su = sum(v["avg_us"] * v["launches_per_sequence"] for v in per_kernel.values())
        if au:   # one stream, one context: the sequence's kernel intervals cannot add up to more than the wall time of the probed step
            rl["alone"]["kernel_us_per_sequence"] = su
            rl["alone"]["ok"] = bool(su <= au * 1.02 + 1.0)
    if check_ok is False and not under_profiler:
        rl["overlap_check"]["note"] = ("the two clocks disagree: `achieved` / `frac` do not depend on the contended intervals (they come from "
                                       "the launch sequences that ran alone); see the top-level key roofline_check_failed")
    if not pmc_ok:
        return rl
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        wl = pmc.get(workload + "_tick" if tick_key else workload, {})
        if ticks_per_launch == 1 and wl.get("ticks_per_launch", 1) != 1:
            wl = pmc.get(workload + "_tick", {})   # the one-launch tick has its own counter passes
        tpl = wl.get("ticks_per_launch", 1)
        scale = ticks_per_launch / float(tpl)      # the counters are per launch of `tpl` ticks; per tick they do not depend on it
        ents = {k: v for k, v in wl.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}
        if ents:
            by_stage = {k: int(v["hbm_bytes_per_launch"] * scale) for k, v in ents.items()}
            rl["traffic"] = int(sum(by_stage.values()))
            rl["traffic_by_kernel"] = by_stage
            rl["traffic_over_required"] = rl["traffic"] / b_req if b_req else None
            rl["traffic_over_algorithmic"] = rl["traffic"] / b_launch if b_launch else None
            rl["traffic_from_profile_file"] = True
            rl["traffic_source"] = ("%s (commit %s; all kernels of a launch sequence, per launch of %d ticks, scaled to %d): rocprofv3 "
                                    "--pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, (factor*FETCH_SIZE + WRITE_SIZE)*1024, factor "
                                    "per access pattern from profiles/fetch_calibration.json"
                                    % (wl.get("csv", "profiles/pmc_traffic.json"), wl.get("commit", "unrecorded"), tpl, ticks_per_launch))
            issue = {k: v["valu_issue_us_per_launch"] * scale for k, v in ents.items() if "valu_issue_us_per_launch" in v}
            if issue:
                rl["valu_issue"] = {"from_profile_file": True, "commit": wl.get("commit", "unrecorded"),
                                    "kernel_issue_us_per_launch": issue, "sum_us_per_launch": sum(issue.values()),
                                    "chip_utilisation": sum(issue.values()) / (step_s * 1e6),
                                    "lane_utilisation": {k: v.get("valu_lane_utilisation") for k, v in ents.items()
                                                         if v.get("valu_lane_utilisation") is not None} or None,
                                    "what": "SQ_ACTIVE_INST_VALU (quad-cycles over 1024 SIMDs at 2.4 GHz) of every kernel of one launch "
                                            "sequence, summed, over THIS run's driver-timed step: the share of the chip's vector issue "
                                            "slots the step uses"}
                rl["step"]["valu_frac"] = rl["valu_issue"]["chip_utilisation"]
                if rl["step"]["valu_frac"] > rl["step"]["hbm_frac"]:
                    rl["bound"] = "valu"
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return rl


def probe_pass(eng, sync, calls):
    """The same launch sequences ALONE on the device (one context, nothing else in flight): once without probes, once with
    every launch probed -- the kernels' own intervals for the roofline line, and what a probe costs a launch in this run.
    `calls`: 2k prepared steps (the first k run unprobed, the last k probed); sync(): wait for the context's stream."""
    k = len(calls) // 2
    if k < 1:
        return None
    sync()
    t0 = time.perf_counter()
    for c in calls[:k]:
        c()
    sync()
    t_plain = (time.perf_counter() - t0) / k
    eng.profile_enable(1)
    t0 = time.perf_counter()
    for c in calls[k:2 * k]:
        c()
    sync()
    t_probed = (time.perf_counter() - t0) / k
    n_samples, _ = eng.profile_read()
    kernels = {name: {"launches": l, "ms": ms, "stage": st} for name, (l, ms, st) in eng.profile_kernels().items()}
    eng.profile_enable(0)
    launches = sum(v["launches"] for v in kernels.values()) / max(1, n_samples)
    return {"kernels": kernels, "n_samples": n_samples, "sequences": k, "step_us_plain": t_plain * 1e6, "step_us_probed": t_probed * 1e6,
            "launches_per_sequence": launches,
            "probe_cost_us_per_launch": max(0.0, (t_probed - t_plain) * 1e6 / max(1.0, launches))}


def cpu_baseline(wl, nodes, sources, cpu_ticks):
    """The CPU oracle (a port of the reference loop; the reference is Java and no JVM exists on
    the box) timed on a bounded sample of the same workload: one thread, as the reference
    evaluates a packet on one thread (SimulatorJSONHandler.java:93)."""
    from oracle import oracle as O
    from radio_sim_amd import workload as W
    idx, n, frac, model, _ = WORKLOADS[wl]
    kind, kw = W.model_kwargs(model)
    okind = {"udgm": O.MODEL_UDGM, "udgm_const": O.MODEL_UDGM_CONST, "logdist": O.MODEL_LOGDIST}[kind]
    okw = {("ld_flags" if k == "flags" else k): v for k, v in kw.items()}
    mdl = O.model(okind, **okw)
    nd = O.NodeTable(n)
    nd.x, nd.y, nd.z = nodes.x, nodes.y, nodes.z
    t = len(sources[0])
    n_pk = max(1, int(round(cpu_ticks * t)))
    src = np.concatenate(sources[: (n_pk + t - 1) // t])[:n_pk]
    pk = nd.packets(src, 0, W.AIR_US)
    O.count_links(mdl, nd, pk[:2], 0, 1)     # page in
    t0 = time.perf_counter()
    heard, deliv = O.count_links(mdl, nd, pk, 0, 1)
    dt = time.perf_counter() - t0
    links = n_pk * (n - 1)
    out = {"value": links / dt, "unit": "links/s", "cores": 1, "kind": "port",
           "sample": "%d packets x %d receivers (%.1f ticks of this workload), %.1f s, oracle/rm_oracle.c "
                     "single thread" % (n_pk, n - 1, n_pk / t, dt)}
    threads = O.lib().orc_max_threads()
    t0 = time.perf_counter()
    O.count_links(mdl, nd, pk, 0, threads)
    dt_mt = time.perf_counter() - t0
    mt = {"value": links / dt_mt, "unit": "links/s", "cores": threads, "kind": "port",
          "sample": "same sample, OpenMP over packets, %.1f s" % dt_mt}
    return out, mt


def scale_probe(rsa, W, torch, dev, device_ordinal, inflight, batch, ticks=384, warm=96):
    """The same medium at 1M nodes / 1000 frames per tick (38 MB of algorithmic traffic per tick): where
    the sweep stops being launch-latency-bound.  Same measurement rules as the main run."""
    idx, n, frac, model, desc = WORKLOADS["m1"]
    t_per_tick = int(round(frac * n))
    nodes = W.make_nodes(n, idx)
    kind_name, kw = W.model_kwargs(model)
    engines, streams = [], []
    for _ in range(inflight):
        e = rsa.Engine(device_ordinal)
        st = torch.cuda.Stream(device=dev)
        e.set_stream(st.cuda_stream)
        e.upload_table(nodes)
        e.set_model(rsa.MODEL_LOGDIST, **kw)
        e.set_link_capacity(1 << 21)
        engines.append(e)
        streams.append(st)
    sources = [W.choose_sources(n, t_per_tick, 0xC0FFEE00 + idx, k) for k in range(warm + ticks)]
    with torch.cuda.stream(streams[0]):
        src_dev = torch.from_numpy(np.stack(sources)).to(dev)
    streams[0].synchronize()

    batch = max(1, min(batch, 16))  # 16 ticks of this size already fill the device

    def run(k0, k1):
        g = 0
        for k in range(k0, k1, batch):
            nb = min(batch, k1 - k)
            t0 = np.arange(k, k + nb, dtype=np.int64) * W.TICK_US
            ptrs = np.array([src_dev[kk].data_ptr() for kk in range(k, k + nb)], dtype=np.uint64)
            if batch == 1:
                engines[g % inflight].tick_run_sources_device(int(t0[0]), int(t0[0]) + W.TICK_US, int(ptrs[0]), t_per_tick,
                                                              int(t0[0]), W.AIR_US)
            else:
                engines[g % inflight].batch_run_sources_device(t0, t0 + W.TICK_US, ptrs, np.full(nb, t_per_tick, dtype=np.int32),
                                                               t0, np.full(nb, W.AIR_US, dtype=np.int64))
            g += 1

    def fence():
        for st in streams:
            st.synchronize()
        torch.cuda.synchronize()

    run(0, warm)
    fence()
    engines[0].profile_enable(4)
    t0 = time.perf_counter()
    run(warm, warm + ticks)
    fence()
    el = time.perf_counter() - t0
    n_samples, _ = engines[0].profile_read()
    kernels = {name: {"launches": l, "ms": ms, "stage": st} for name, (l, ms, st) in engines[0].profile_kernels().items()}
    engines[0].profile_enable(0)
    heard, dropped = engines[0].result_count()
    try:
        cand, _ = engines[0].slot_stats(0)
    except Exception:
        cand = 0
    per_tick = el / ticks
    alone = None
    if batch > 1:   # the same launch sequences alone on the device: the kernels' own intervals
        t0 = np.arange(batch, dtype=np.int64) * W.TICK_US
        ptrs = np.array([src_dev[kk].data_ptr() for kk in range(batch)], dtype=np.uint64)
        call = engines[0].prepared("rm_batch_run_sources_device", batch, t0, t0 + W.TICK_US, ptrs, np.full(batch, t_per_tick, dtype=np.int32),
                                   t0, np.full(batch, W.AIR_US, dtype=np.int64))
        alone = probe_pass(engines[0], streams[0].synchronize, [call] * 8)
    rl = roofline_object(kernels=kernels, n_samples=n_samples, n_loc=n, t_per_tick=t_per_tick, heard=heard, cand=cand,
                         ticks_per_launch=batch, step_s=per_tick * batch, contexts=inflight, workload="m1", alone=alone,
                         tile_reuse=engines[0].batch_tile_reuse() if batch > 1 else 1)
    for e in engines:
        e.close()
    return {"workload": desc, "nodes": n, "tx_per_tick": t_per_tick, "ticks_in_flight": inflight * batch,
            "ticks_per_launch": batch, "contexts": inflight,
            "value": t_per_tick * (n - 1) / per_tick, "unit": "links/s", "ms_per_tick": per_tick * 1e3,
            "ms_per_step": per_tick * batch * 1e3, "roofline": rl, "dropped": bool(dropped)}


def host_transfer_legs(rsa, W, eng, stream, torch, nodes, sources, n, t_per_tick, tick_us, src_dev, pool, batch):
    """The same ticks with the data crossing PCIe, driver-timed like everything else in this file (SURVEY.md 8(d):
    "report separately with/without host<->device transfers"; never the headline `value`):
      tick_flush_view    the closed loop a JNI tick mode sees: records in (rm_enqueue_tx_records), ONE evaluation, the
                         heard links read in place from the context's pinned block (rm_tick_flush_view)
      tick_events        the same loop with the reception stage on the device: nothing but the drain's deliveries
                         crosses the link (rm_tick_run + rm_events_process)
      batch_result_view  `batch` ticks per launch sequence, all their records brought to the host in one packing launch"""
    import ctypes as C
    out = {}
    recs = []
    for s in sources[:16]:
        r = np.zeros(len(s), dtype=rsa.TX_RECORD_DTYPE)
        r["x"], r["y"], r["z"] = nodes.x[s], nodes.y[s], nodes.z[s]
        r["txpower"], r["txprob"], r["channel"] = nodes.txpower[s], nodes.txprob[s], nodes.channel[s]
        r["src"], r["air_us"] = s, W.AIR_US
        recs.append(r)
    links_per_tick = t_per_tick * (n - 1)

    # the C ABI called directly (what a compiled host does: tools/loop_latency.cpp is the same loop in C++); the binding's
    # conveniences -- array wrappers, argument conversion -- are not part of the engine
    from radio_sim_amd._lib import HostResult, DeliveryView, check
    L, h = eng._L, eng._h
    rec_ptr = [C.c_void_p(r.ctypes.data) for r in recs]
    n_rec = len(recs[0])
    hres = HostResult()

    def flush_loop(k0, k1):
        got = 0
        for k in range(k0, k1):
            check(L.rm_tick_begin(h, k * tick_us, (k + 1) * tick_us))
            check(L.rm_enqueue_tx_records(h, rec_ptr[k % len(recs)], n_rec))
            check(L.rm_tick_flush_view(h, C.byref(hres)))
            got += hres.count
        return got
    with torch.cuda.stream(stream):
        flush_loop(0, 16)
        reps = 200
        t0 = time.perf_counter()
        got = flush_loop(16, 16 + reps)
        el = time.perf_counter() - t0
    out["tick_flush_view"] = {"us_per_tick": el / reps * 1e6, "value": links_per_tick * reps / el, "unit": "links/s",
                              "bytes_out_per_tick": got / reps * 13 + t_per_tick * 5, "bytes_in_per_tick": t_per_tick * 64}
    # the reception stage on: frames of 8128 us over 1000 us ticks, ~9 ticks of packets pending at any time
    with torch.cuda.stream(stream):
        eng.events_enable(1 << 16, 1 << 21)
        eng.set_time(0)

        split = [0.0, 0.0]   # host time inside the two calls (the drain call includes waiting for the device)

        src_ptr = [C.c_void_p(src_dev[k].data_ptr()) for k in range(pool)]
        dview = DeliveryView()

        def ev_loop(k0, k1):
            got = 0
            for k in range(k0, k1):
                ta = time.perf_counter()
                check(L.rm_tick_run_sources_device(h, k * tick_us, (k + 1) * tick_us, src_ptr[k % pool], t_per_tick, k * tick_us, W.AIR_US))
                tb = time.perf_counter()
                check(L.rm_events_process(h, (k + 1) * tick_us, C.byref(dview)))   # the deliveries are read in place
                got += dview.count
                split[0] += tb - ta
                split[1] += time.perf_counter() - tb
            return got
        ev_loop(0, 24)
        split[:] = [0.0, 0.0]
        t0 = time.perf_counter()
        got = ev_loop(24, 24 + reps)
        el = time.perf_counter() - t0
        eng.events_disable()
    out["tick_events"] = {"us_per_tick": el / reps * 1e6, "value": links_per_tick * reps / el, "unit": "links/s",
                          "deliveries_per_tick": got / reps, "bytes_out_per_tick": got / reps * 12 + t_per_tick * 16,
                          "tick_call_us": split[0] / reps * 1e6, "drain_call_us": split[1] / reps * 1e6,
                          "what": "rm_tick_run_sources_device + rm_events_process per tick: Simulator.generate*Events, "
                                  "processAllEvents and the Transciever state on the device, the deliveries on the host"}
    if batch > 1:
        nb = min(batch, pool)
        t_b = np.arange(nb, dtype=np.int64) * tick_us
        ptrs = np.array([src_dev[k].data_ptr() for k in range(nb)], dtype=np.uint64)
        cnt = np.full(nb, t_per_tick, dtype=np.int32)
        air = np.full(nb, W.AIR_US, dtype=np.int64)
        with torch.cuda.stream(stream):
            for _ in range(2):
                eng.batch_run_sources_device(t_b, t_b + tick_us, ptrs, cnt, t_b, air)
                views, _ = eng.batch_result_view(nb)
            reps_b = 8
            t0 = time.perf_counter()
            for _ in range(reps_b):
                eng.batch_run_sources_device(t_b, t_b + tick_us, ptrs, cnt, t_b, air)
                views, _ = eng.batch_result_view(nb)
            el = time.perf_counter() - t0
        links = sum(v.count for v in views)
        out["batch_result_view"] = {"us_per_tick": el / (reps_b * nb) * 1e6, "value": links_per_tick * reps_b * nb / el,
                                    "unit": "links/s", "ticks_per_launch": nb, "bytes_out_per_tick": links / nb * 13}
        # the same with the reference's own UDGM medium: a heard link's rssi is its packet's transmit power (UDGMRadioMedium.java:95),
        # so ONE value per packet crosses the link and a link is 5 bytes (node, verdict) instead of 13 (ABI version 5)
        with torch.cuda.stream(stream):
            eng.set_model(rsa.MODEL_UDGM)
            for _ in range(2):
                eng.batch_run_sources_device(t_b, t_b + tick_us, ptrs, cnt, t_b, air)
                views, _ = eng.batch_result_view(nb)
            t0 = time.perf_counter()
            for _ in range(reps_b):
                eng.batch_run_sources_device(t_b, t_b + tick_us, ptrs, cnt, t_b, air)
                views, _ = eng.batch_result_view(nb)
            el = time.perf_counter() - t0
        links = sum(v.count for v in views)
        out["batch_result_view_udgm"] = {"us_per_tick": el / (reps_b * nb) * 1e6, "value": links_per_tick * reps_b * nb / el, "unit": "links/s",
                                         "ticks_per_launch": nb, "bytes_out_per_tick": links / nb * 5 + t_per_tick * 13,
                                         "what": "reference UDGM, rssi once per packet: rm_host_result.rssi is NULL, pkt_rssi carries it"}
    return out


def dense_probe(rsa, W, torch, dev, device_ordinal, n=20_000, t=200, ticks=24):
    """Layouts the spatial cull cannot help: 20k nodes, 200 frames per tick, every frame heard by every node -- the reference
    UDGM medium with a range beyond the square's diagonal, and the reference's DEFAULT medium (NullRadioMedium.java:47-77:
    every same-channel node hears everything).  4 M heard links per tick are evaluated in node order (rm_dense.hip), one tick at a
    time.  Two legs per medium: the tick as it ends by default -- ONE launch, the heard links as lane masks per (frame, 1024 nodes)
    cell, which IS the result of such a medium (rm_result_dense: a link's rssi and verdict are its packet's; offsets and totals
    are laid out when a reader asks) -- and, `_records`, the same tick
    with its 17-byte records written out every time (RM_DENSE_LAZY=0: 68 MB of record writes per tick)."""
    out = {}
    nodes = W.make_nodes(n, 3)
    side = W.side_length(n)
    chunks = -(-n // 1024)
    for name, kind, kw in (("udgm_everyone_in_range", rsa.MODEL_UDGM, dict(udgm_transmission_range=float(side * 1.5))),
                           ("null_medium", rsa.MODEL_NULL, {})):
        e = rsa.Engine(device_ordinal)
        st = torch.cuda.Stream(device=dev)
        e.set_stream(st.cuda_stream)
        e.upload_table(nodes)
        e.set_model(kind, **kw)
        e.set_link_capacity(1 << 24)
        srcs = [W.choose_sources(n, t, 0xC0FFEE0D, k) for k in range(4)]
        with torch.cuda.stream(st):
            src_dev = torch.from_numpy(np.stack(srcs)).to(dev)
            st.synchronize()
            ticks_only = [e.prepared("rm_tick_run_sources_device", k * 1000, k * 1000 + 1000, ctypes.c_void_p(src_dev[k % 4].data_ptr()), t, k * 1000,
                                     W.AIR_US) for k in range(ticks)]   # (the arguments converted once: the loop times the engine, not the binding)
            for leg, calls in (("", ticks_only), ("_records", ticks_only)):
                # (the records at once: RM_DENSE_LAZY=0 is read per tick -- the write pass is then one of the tick's own launches)
                if leg:
                    os.environ["RM_DENSE_LAZY"] = "0"
                else:
                    os.environ.pop("RM_DENSE_LAZY", None)
                for k in range(3):
                    calls[k]()
                heard, dropped = e.result_count()
                st.synchronize()
                e.profile_enable(4)
                t0 = time.perf_counter()
                for call in calls:
                    call()
                st.synchronize()
                el = time.perf_counter() - t0
                heard, dropped = e.result_count()
                n_samples, _ = e.profile_read()
                kernels = {nm: {"launches": l, "ms": ms, "stage": sg} for nm, (l, ms, sg) in e.profile_kernels().items()}
                e.profile_enable(0)
                alone = probe_pass(e, st.synchronize, calls[:16])
                per_tick = el / ticks
                if leg == "":   # node columns in, lane masks + counts out
                    req = (n * S_NODE + t * S_TX + t * chunks * (16 * 8 + 4),
                           "N*37 + T*56 in, T x %d cells x (16 lane masks + a count) out: the dense tick's own result (rm_result_dense lays "
                           "the cells out -- packet offsets, totals -- when it is called; the records are not written)" % chunks)
                else:
                    req = (n * S_NODE + t * S_TX + heard * (S_REC - 8), "N*37 + T*56 + H*17: 8(d) with 17-byte records (no sinr column)")
                rl = roofline_object(kernels=kernels, n_samples=n_samples, n_loc=n, t_per_tick=t, heard=heard, cand=0, ticks_per_launch=1,
                                     step_s=per_tick, contexts=1, workload="dense_" + name, pmc_ok=False, alone=alone, required=req)
                if leg == "_records":
                    try:    # the counter passes on file for this shape (the count pass per medium, the write pass of both)
                        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("dense", {})
                        parts = [pmc.get("k_dense_count_" + ("null" if name == "null_medium" else "udgm")), pmc.get("k_dense_write")]
                        if all(isinstance(v, dict) and "hbm_bytes_per_launch" in v for v in parts):
                            rl["traffic"] = int(sum(v["hbm_bytes_per_launch"] for v in parts))
                            rl["traffic_over_required"] = rl["traffic"] / rl["required_bytes_per_launch"]
                            rl["traffic_source"] = ("profiles/pmc_traffic.json ('dense', commit %s): FETCH_SIZE / WRITE_SIZE passes of bench.py "
                                                    "--dense-only, count + write kernels of one tick; from_profile_file" % pmc.get("commit"))
                    except (OSError, ValueError):
                        pass
                out[name + leg] = {"workload": "20k nodes, 200 frames per tick, every frame heard by every node (nothing to cull): " + name +
                                               (" -- records written every tick" if leg else " -- lane masks (rm_result_dense)"),
                                   "nodes": n, "tx_per_tick": t, "heard_links_per_tick": int(heard), "ms_per_tick": per_tick * 1e3,
                                   "value": t * (n - 1) / per_tick, "unit": "links/s",
                                   "record_bytes_per_s": (heard * (S_REC - 8) / per_tick) if leg else None, "roofline": rl, "dropped": bool(dropped)}
            os.environ.pop("RM_DENSE_LAZY", None)
        e.close()
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks itself -- fresh child processes
    (python -m torch.distributed.run, one rank per GPU over RCCL) -- BEFORE it has imported torch or touched a GPU, relays
    rank 0's JSON line and exits with the children's code.  Nothing that has initialised the GPU is ever re-executed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    out, _ = proc.communicate()
    line = None
    for ln in out.decode(errors="replace").splitlines():      # rank 0 prints exactly one JSON line; the launcher may add its own
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    raise SystemExit(proc.returncode if proc.returncode else (0 if line is not None else 1))


def dry_run(args, rank, world, result_fd):
    """RM_BENCH_DRY_RUN=1: the launch plumbing without a GPU (CPU test tier): the ranks rendezvous over gloo, agree on a
    number, rank 0 prints one line."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        total = 1.0
    if rank == 0:
        os.write(result_fd, (json.dumps({"metric": baseline_metric(), "dry_run": True, "n_gpus": world, "steps": args.steps,
                                         "warmup": args.warmup, "rank_sum": total}) + "\n").encode())


# ---------------------------------------------------------------------------------------------------------------------
# One configuration through the bench.  `measure()` is the path the driver runs (`python bench.py [--gpus N --steps K
# --warmup W]`), top to bottom: set-up, input, the mode's issue function, the timed region, the result line.  Every mode is one
# function that returns `issue(k0, k1, plan_only)` -- "put ticks k0 .. k1-1 on the device":
#     mode_batches          rm_batch_run_sources_device, `inflight` contexts in turn -- ONE GPU, THE DRIVER'S DEFAULT
#     mode_as_rank          --as-rank R:W: one rank's share behind the call a real rank makes behind its all-gather
#     mode_lib_dist         several GPUs: rm_dist_batch_run_sources_device (stage + ncclAllGather + sweep, one call per batch)
#     mode_torch_batches    several GPUs / --force-sharded: torch.distributed runs the collective around the engine calls
#     mode_torch_ticks      ... one tick at a time (draws, frames that stay on the air)
#     mode_lone_ticks       --batch 1: rm_tick_run_sources_device per tick

class Env:
    """what the process is: rank, world, device, the (optional) process group"""


class Run:
    """the state of one configuration: workload, engines, input, clock"""


def setup(args, env):
    """workload -> nodes, medium, `inflight` engine contexts (each with its own stream), the receiver partition"""
    rsa, W, torch = env.rsa, env.W, env.torch
    run = Run()
    run.args, run.env = args, env
    idx, n, frac, model, desc = WORKLOADS[args.workload]
    run.as_rank = tuple(int(v) for v in args.as_rank.split(":")) if args.as_rank else None
    if run.as_rank and args.scaling == "weak":
        n = int(round(n * run.as_rank[1] ** 0.5))
        desc += " -- compute of rank %d of %d (weak scaling: %d nodes), no collective" % (run.as_rank[0], run.as_rank[1], n)
    elif run.as_rank:
        desc += " -- compute of rank %d of %d (strong scaling: its share of the %d receivers), no collective" % (run.as_rank[0], run.as_rank[1], n)
    elif args.nodes > 0:
        n = args.nodes
        desc += " -- node count overridden: %d" % n
    elif env.world > 1 and args.scaling == "weak":
        # per-GPU link evaluations per tick fixed: T x N_loc = f*N * N/world = const  =>  N ~ sqrt(world)
        n = int(round(n * env.world ** 0.5))
        desc += " -- weak scaling: %d nodes on %d GPUs, same density and Tx fraction" % (n, env.world)
    run.idx, run.n, run.model, run.desc = idx, n, model, desc
    run.t_per_tick = int(round(frac * n))
    extra = dict(EXTRA.get(args.workload, {}))
    if args.link_capacity > 0:
        extra["link_capacity"] = args.link_capacity
    run.extra = extra
    run.tick_us = extra.get("tick_us", W.TICK_US)
    # the SINR extension looks at every frame on the air: ticks are chained through the frames on the air unless no frame
    # outlives its tick -- ONE context, one timeline; a batch of such ticks is still one launch sequence (rm_airbatch.hip)
    run.stateful = model in ("logdist_sinr16", "logdist_sinr_overlap") and W.AIR_US > run.tick_us
    run.sinr_column = model in ("logdist_sinr16", "logdist_sinr_overlap")
    run.nodes = W.make_nodes(n, idx, channels16=extra.get("channels16", False))
    if args.spatial_ids:
        # Morton order of the positions: index ranges become spatial regions (what a host that assigns node ids by
        # location gives the range partition of the multi-GPU mode)
        nodes = run.nodes
        side = float(max(nodes.x.max(), nodes.y.max())) + 1e-9
        qx = np.minimum((nodes.x / side * 65536).astype(np.uint64), 65535)
        qy = np.minimum((nodes.y / side * 65536).astype(np.uint64), 65535)

        def spread(v):
            v = (v | (v << 8)) & np.uint64(0x00FF00FF)
            v = (v | (v << 4)) & np.uint64(0x0F0F0F0F)
            v = (v | (v << 2)) & np.uint64(0x33333333)
            return (v | (v << 1)) & np.uint64(0x55555555)
        order = np.argsort(spread(qx) | (spread(qy) << np.uint64(1)), kind="stable")
        for f in ("x", "y", "z", "txpower", "channel", "enabled", "rxprob", "txprob"):
            setattr(nodes, f, np.ascontiguousarray(getattr(nodes, f)[order]))
        run.desc += " -- node ids along a Morton curve (--spatial-ids)"
    kind_name, kw = W.model_kwargs(model)
    kind = {"udgm": rsa.MODEL_UDGM, "udgm_const": rsa.MODEL_UDGM_CONST, "logdist": rsa.MODEL_LOGDIST}[kind_name]
    run.inflight = max(1, args.inflight) if not run.stateful else 1
    run.engines, run.streams = [], []
    share = run.as_rank[1] if run.as_rank else env.world
    for _ in range(run.inflight):
        e = rsa.Engine(env.device_ordinal)
        st = torch.cuda.Stream(device=env.dev)
        e.set_stream(st.cuda_stream)
        e.upload_table(run.nodes)
        e.set_model(kind, **kw)
        # (a rank's share of the links: its result slots -- up to 512 per context -- are sized for it, with a margin)
        e.set_link_capacity(max(1 << 17, extra.get("link_capacity", 1 << 21) * 2 // share) if share > 1 else extra.get("link_capacity", 1 << 21))
        run.engines.append(e)
        run.streams.append(st)
    run.eng, run.stream = run.engines[0], run.streams[0]
    # receiver partitioning (strong scaling): rank r owns a region of the plane, or the receivers [lo, hi)
    run.spatial = args.partition == "spatial"
    run.part_r, run.part_w = (run.as_rank if run.as_rank else (env.rank, env.world))
    run.lo = (n * run.part_r) // run.part_w
    run.hi = (n * (run.part_r + 1)) // run.part_w
    run.own = None                          # owner of every node (who packs a transmitter's record)
    if run.part_w > 1:
        from radio_sim_amd import dist as D0
        run.own = D0.owners(n, run.part_w, run.eng if run.spatial else None)
    run.n_loc = int((run.own == run.part_r).sum()) if run.own is not None else n
    if run.as_rank:
        for e in run.engines:
            if run.spatial:
                e.set_partition_spatial(run.part_r, run.part_w)
            else:
                e.set_partition(run.lo, run.hi - run.lo)
        run.desc += " (%s partition: %d receivers)" % (args.partition, run.n_loc)
    # a step is one launch sequence of `batch` ticks
    run.batch = max(1, min(extra.get("batch", args.batch) if (args.batch_default and env.world == 1 and not run.as_rank) else args.batch, rsa.MAX_BATCH))
    run.steps = args.steps if args.steps > 0 else -(-1920 // run.batch)
    run.warmup = args.warmup if args.warmup >= 0 else -(-192 // run.batch)
    run.warm_ticks, run.ticks = run.warmup * run.batch, (run.warmup + run.steps) * run.batch
    # synthetic input: a pool of distinct ticks (a multiple of the batch), reused in turn by longer runs
    run.pool = min(run.ticks, -(-2112 // run.batch) * run.batch)
    run.sources = [W.choose_sources(n, run.t_per_tick, 0xC0FFEE00 + idx, k) for k in range(run.pool)]
    run.clock = 0          # simulated ticks issued so far: simulated time never runs backwards (the SINR medium keeps frames on the air)
    run.links_done = 0
    run.last_run = (run.eng, 0)   # (context, result slot) of the last tick issued
    run.ctx_rr = 0
    run.prepared, run.keep_alive = {}, []
    return run


def join_library_collective(run):
    """several GPUs, --collective lib: one RCCL communicator per context inside libradiomedium_hip.so (rank 0 makes the ids,
    torch.distributed -- only the out-of-band channel here -- hands them round).  False: torch runs the collective."""
    args, env = run.args, run.env
    rsa, torch, dist = env.rsa, env.torch, env.dist
    ok = 1 if rsa.Engine.comm_available() else 0
    if env.world > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device=env.dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if ok and args.collective != "lib" and os.environ.get("RM_BENCH_NO_LIB_COLLECTIVE") == "1":
        ok = 0
    if not ok:
        if args.collective == "lib":
            raise SystemExit("--collective lib: RCCL could not be bound inside libradiomedium_hip.so on every rank")
        print("bench: the library could not bind RCCL on every rank: torch.distributed runs the collective", file=sys.stderr)
        return False
    joined = 1
    for e in run.engines:
        if env.world > 1:
            e.set_partition_spatial(env.rank, env.world) if run.spatial else e.set_partition(run.lo, run.hi - run.lo)
        uid = torch.from_numpy(rsa.Engine.comm_unique_id() if env.rank == 0 else np.zeros(128, dtype=np.uint8))
        if env.world > 1:
            uid = uid.to(env.dev)
            dist.broadcast(uid, src=0)
            uid = uid.cpu()
        try:
            e.comm_init_rank(uid.numpy(), env.world, env.rank)
        except rsa.RadioMediumError as err:     # (reported, and every rank then takes torch's collective together)
            print("bench: rank %d could not join the library's communicator: %s" % (env.rank, err), file=sys.stderr)
            joined = 0
            break
    if env.world > 1:
        flag = torch.tensor([joined], dtype=torch.int32, device=env.dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        joined = int(flag.item())
    if not joined:
        if args.collective == "lib":
            raise SystemExit("--collective lib: a rank could not join the library's communicator")
        for e in run.engines:
            e.comm_destroy()
    return bool(joined)


def step_times(run, nb, tk):
    return np.arange(tk, tk + nb, dtype=np.int64) * run.tick_us


def issue_steps(run, step_call):
    """the issue function of the batched single-process modes: contexts take the batches in turn; every step is a prepared call
    (its arguments converted once: what the interpreter adds per step would otherwise bound a rank whose share of a batch
    needs less device time than the argument conversion takes)"""
    def issue(k0, k1, plan_only=False):
        with run.env.torch.cuda.stream(run.stream):
            rr = run.ctx_rr
            for k in range(k0, k1, run.batch):
                nb = min(run.batch, k1 - k)
                g = rr % run.inflight
                rr += 1
                call = step_call(g, k, nb, run.clock + k - k0)
                if not plan_only:
                    call()
                    run.last_run = (run.engines[g], nb - 1)
            if plan_only:
                return
            run.ctx_rr = rr
        run.clock += k1 - k0
    return issue


def cached(run, key, make):
    if key not in run.prepared:
        run.prepared[key] = make()
    return run.prepared[key]


def mode_batches(run):
    """ONE GPU, the driver's default: `batch` ticks per rm_batch_run_sources_device call, the source lists resident in HBM"""
    W, torch = run.env.W, run.env.torch
    with torch.cuda.stream(run.stream):
        run.src_dev = torch.from_numpy(np.stack(run.sources)).to(run.env.dev)                    # [ticks, T] int32
    run.stream.synchronize()

    def step_call(g, k, nb, tk):
        def make():
            t0 = step_times(run, nb, tk)
            ptrs = np.array([run.src_dev[kk % run.pool].data_ptr() for kk in range(k, k + nb)], dtype=np.uint64)
            return run.engines[g].prepared("rm_batch_run_sources_device", nb, t0, t0 + run.tick_us, ptrs, np.full(nb, run.t_per_tick, dtype=np.int32),
                                           t0, np.full(nb, W.AIR_US, dtype=np.int64))
        return cached(run, (g, k, nb, tk), make)
    run.step_call = step_call
    return issue_steps(run, step_call)


def padded_sources(run, ranks):
    """every rank's transmitters of every tick in a fixed number of slots (src = -1: padding), on the device"""
    from radio_sim_amd import dist as D
    torch = run.env.torch
    run.slots = D.slots_needed(run.n, run.part_w, run.sources, run.own)
    return {r: torch.from_numpy(np.stack([D.pad_sources(s[run.own[s] == r] if run.own is not None else s, run.slots)
                                          for s in run.sources])).to(run.env.dev) for r in ranks}


def mode_as_rank(run):
    """--as-rank R:W on one GPU: what the all-gather of the source indices would deliver, [rank][tick][slot]; the call builds
    every rank's records from the node table, as a rank of a real run does behind its collective"""
    W, torch, args = run.env.W, run.env.torch, run.args
    with torch.cuda.stream(run.stream):
        run.src_dev = torch.from_numpy(np.stack(run.sources)).to(run.env.dev)
        if args.host_cull > 0:
            nodes, mine, m = run.nodes, run.eng.partition_nodes(), args.host_cull
            bx0, bx1, by0, by1 = nodes.x[mine].min(), nodes.x[mine].max(), nodes.y[mine].min(), nodes.y[mine].max()
            run.sources = [s[(nodes.x[s] >= bx0 - m) & (nodes.x[s] <= bx1 + m) & (nodes.y[s] >= by0 - m) & (nodes.y[s] <= by1 + m)]
                           for s in run.sources]
            run.desc += " -- HOST-CULLED frames (experiment): %.0f of %d per tick" % (np.mean([len(s) for s in run.sources]), run.t_per_tick)
        pad_dev = padded_sources(run, range(run.part_w))
    run.stream.synchronize()

    def step_call(g, k, nb, tk):
        def make():
            t0 = step_times(run, nb, tk)
            k_in = k % run.pool                      # (the pool is a multiple of the batch: the window does not wrap)
            gathered = torch.stack([pad_dev[r][k_in:k_in + nb] for r in range(run.part_w)]).contiguous()
            run.keep_alive.append(gathered)
            return run.engines[g].prepared("rm_batch_run_gathered_sources_device", nb, t0, t0 + run.tick_us, ctypes.c_void_p(gathered.data_ptr()),
                                           run.part_w, run.slots, t0, W.AIR_US)
        return cached(run, (g, k, nb, tk), make)
    run.step_call = step_call
    return issue_steps(run, step_call)


def mode_lib_dist(run):
    """several GPUs: per batch ONE call -- this rank's block (source indices + the node table's digest) staged, ncclAllGather
    inside the library on the context's stream, the frame list, the sweep"""
    W, torch = run.env.W, run.env.torch
    with torch.cuda.stream(run.stream):
        run.src_dev = torch.from_numpy(np.stack(run.sources)).to(run.env.dev)
        pad_dev = padded_sources(run, [run.env.rank])
    run.stream.synchronize()

    def step_call(g, k, nb, tk):
        def make():
            t0 = step_times(run, nb, tk)
            return run.engines[g].prepared("rm_dist_batch_run_sources_device", nb, t0, t0 + run.tick_us,
                                           ctypes.c_void_p(pad_dev[run.env.rank][k % run.pool].data_ptr()), run.slots, t0, W.AIR_US)
        return cached(run, (g, k, nb, tk), make)
    run.step_call = step_call
    return issue_steps(run, step_call)


def make_sharded_driver(run):
    """torch.distributed around the engine calls (radio-sim_amd/dist.py: ShardedTick); also the gloo rehearsal on one GPU"""
    from radio_sim_amd import dist as D
    env, torch = run.env, run.env.torch
    with torch.cuda.stream(run.stream):
        # every rank packs the frames whose source it owns into a fixed number of slots (padded with src = -1), then the
        # ranks all-gather the slots over RCCL
        run.slots = D.slots_needed(run.n, env.world, run.sources, run.own)
        pad = np.stack([D.pad_sources(s[run.own[s] == env.rank] if run.own is not None else s, run.slots) for s in run.sources])
        run.src_dev = torch.from_numpy(pad).to(env.dev)
        run.sharded = D.ShardedTick(run.engines, env.dist, run.n, env.rank, env.world, run.slots, env.dev, run.streams, may_draw=False,
                                    batch=run.batch, on_air=run.stateful, spatial=run.spatial)
    run.stream.synchronize()
    return run.sharded


def mode_torch_batches(run):
    """sharded, `batch` ticks per step: packing, one all-gather and the sweep on the context's stream"""
    W = run.env.W
    sharded = make_sharded_driver(run)

    def issue(k0, k1, plan_only=False):
        if plan_only:
            return
        with run.env.torch.cuda.stream(sharded.comm):
            for k in range(k0, k1, run.batch):
                t_b = (run.clock - k0 + np.arange(k, min(k + run.batch, k1), dtype=np.int64)) * run.tick_us
                g = run.ctx_rr % run.inflight
                run.ctx_rr += 1
                sharded.run_batch(g, run.src_dev[k % run.pool].data_ptr(), t_b, W.AIR_US, run.tick_us)
                run.last_run = (run.engines[g], len(t_b) - 1)
        run.clock += k1 - k0
    return issue


def mode_torch_ticks(run):
    """sharded, one tick at a time: the driver prefetches tick k+1 (pack + all-gather) while tick k is swept"""
    W = run.env.W
    sharded = make_sharded_driver(run)

    def issue(k0, k1, plan_only=False):
        if plan_only:
            return
        with run.env.torch.cuda.stream(sharded.comm):
            base = run.clock - k0      # simulated time never runs backwards (frames on the air)
            if k1 > k0:
                sharded.stage(run.src_dev[k0 % run.pool].data_ptr(), (base + k0) * run.tick_us, W.AIR_US)
            for k in range(k0, k1):
                cur = sharded.staged
                if k + 1 < k1:
                    sharded.stage(run.src_dev[(k + 1) % run.pool].data_ptr(), (base + k + 1) * run.tick_us, W.AIR_US)
                sharded.sweep(cur, (base + k) * run.tick_us + run.tick_us)
        run.clock += k1 - k0
    return issue


def mode_lone_ticks(run):
    """--batch 1 on one GPU: one rm_tick_run_sources_device call per tick (the frames' Tx records are built from the resident
    node state inside the sweep)"""
    W, torch = run.env.W, run.env.torch
    with torch.cuda.stream(run.stream):
        run.src_dev = torch.from_numpy(np.stack(run.sources)).to(run.env.dev)
    run.stream.synchronize()

    def issue(k0, k1, plan_only=False):
        if plan_only:
            return
        with torch.cuda.stream(run.stream):
            for k in range(k0, k1):
                t0 = (run.clock + k - k0) * run.tick_us
                run.engines[k % run.inflight].tick_run_sources_device(t0, t0 + run.tick_us, run.src_dev[k % run.pool].data_ptr(), run.t_per_tick,
                                                                      t0, W.AIR_US)
                if run.stateful:
                    run.links_done += run.engines[0].last_link_evaluations()
        run.clock += k1 - k0
    return issue


def pick_mode(run):
    """which issue function this configuration takes"""
    args, env = run.args, run.env
    run.sharded, run.step_call, run.lib_dist = None, None, False
    use_sharded = env.world > 1 or args.force_sharded
    # who runs the collective of a sharded batch: the library (one C call per batch) or torch.distributed around the engine calls
    want_lib = run.batch > 1 and ((env.world > 1 and args.collective != "torch" and env.backend == "nccl")
                                  or (env.world == 1 and args.collective == "lib" and not run.as_rank))
    if want_lib and join_library_collective(run):
        run.lib_dist = True
        return mode_lib_dist(run)
    if run.as_rank and run.batch > 1:
        return mode_as_rank(run)
    if use_sharded:
        return mode_torch_batches(run) if run.batch > 1 else mode_torch_ticks(run)
    return mode_batches(run) if run.batch > 1 else mode_lone_ticks(run)


def fence(run):
    env = run.env
    for st in run.streams:
        st.synchronize()
    env.torch.cuda.synchronize()   # includes the communication stream
    if env.world > 1:
        env.dist.barrier()
        env.torch.cuda.synchronize()


def timed_region(run, issue):
    """set-up passes, W warm-up steps, then EXACTLY K steps between two fences (barrier + synchronize on both sides); returns the
    elapsed time of this rank and the kernel intervals sampled inside the region (for the cross-check only)"""
    # set-up, not a step: every context sweeps one batch once, so that its result slots and link buffers exist before the
    # warm-up (the W warm-up steps alone need not reach every context)
    for _ in range(run.inflight):
        issue(0, min(run.ticks, run.batch))
    fence(run)
    issue(0, run.warm_ticks)
    issue(run.warm_ticks, run.ticks, plan_only=True)     # the timed steps' calls, arguments converted
    fence(run)
    # kernel probes on a few launch sequences of every context: the roofline line takes its kernel times from sequences that run
    # alone after the region; these samples only feed the cross-check, so that the probes cost the headline next to nothing
    every = run.args.profile_every if run.batch == 1 else 4
    if run.steps <= 8 * run.inflight:
        every = 1 if run.steps <= 2 * run.inflight else 2
    for e in run.engines:
        e.profile_enable(every)
    t_start = time.perf_counter()
    run.links_done = 0
    issue(run.warm_ticks, run.ticks)
    fence(run)
    elapsed = time.perf_counter() - t_start
    n_samples, kernels = 0, {}
    for e in run.engines:
        ns, _ = e.profile_read()
        for name, (launches, ms, stage) in e.profile_kernels().items():
            k = kernels.setdefault(name, {"launches": 0, "ms": 0.0, "stage": stage})
            k["launches"] += launches
            k["ms"] += ms
        e.profile_enable(0)
        n_samples += ns
    return elapsed, n_samples, kernels


def alone_pass(run):
    """the same launch sequences ALONE on the device (context 0, nothing else in flight), with and without probes: the kernels'
    own intervals for the roofline line.  Every rank of a sharded run makes the same calls (a collective inside)."""
    if run.sharded is not None:
        return None
    W, torch = run.env.W, run.env.torch
    with torch.cuda.stream(run.stream):
        n_probe = 4 if run.batch > 1 else 64
        if run.batch > 1:
            calls = [run.step_call(0, run.warm_ticks + (j % max(1, run.steps)) * run.batch, run.batch, run.clock + j * run.batch)
                     for j in range(2 * n_probe)]
            run.clock += 2 * n_probe * run.batch
        else:
            calls = lone_tick_calls(run, 2 * n_probe)
        alone = probe_pass(run.eng, run.stream.synchronize, calls)
    fence(run)
    return alone


def lone_tick_calls(run, count):
    """`count` prepared rm_tick_run_sources_device calls at the simulated times that follow"""
    W = run.env.W
    calls = []
    for j in range(count):
        t0 = (run.clock + j) * run.tick_us
        calls.append(run.eng.prepared("rm_tick_run_sources_device", t0, t0 + run.tick_us,
                                      ctypes.c_void_p(run.src_dev[(run.warm_ticks + j) % run.pool].data_ptr()), run.t_per_tick, t0, W.AIR_US))
    run.clock += count
    return calls


def sequential_leg(run, links_per_tick, timed_ticks, heard):
    """`sequential_ticks`: the same ticks again, one at a time on one context -- the closed loop a simulation sees"""
    W, torch, args = run.env.W, run.env.torch, run.args
    fence(run)
    run.eng.profile_enable(args.profile_every)
    t_seq = time.perf_counter()
    with torch.cuda.stream(run.stream):
        seq_ticks = min(timed_ticks, 1920)
        for k in range(run.warm_ticks, run.warm_ticks + seq_ticks):
            t0 = (run.clock + k - run.warm_ticks) * run.tick_us
            run.eng.tick_run_sources_device(t0, t0 + run.tick_us, run.src_dev[k % run.pool].data_ptr(), run.t_per_tick, t0, W.AIR_US)
    fence(run)
    el = time.perf_counter() - t_seq
    seq_samples, _ = run.eng.profile_read()
    seq_kernels = {name: {"launches": l, "ms": ms, "stage": st} for name, (l, ms, st) in run.eng.profile_kernels().items()}
    run.eng.profile_enable(0)
    run.clock += seq_ticks
    with torch.cuda.stream(run.stream):   # the lone tick alone, every launch probed / none probed
        seq_alone = probe_pass(run.eng, run.stream.synchronize, lone_tick_calls(run, 128))
    sequential = {"ticks_in_flight": 1, "ticks": seq_ticks, "value": links_per_tick * seq_ticks / el, "unit": "links/s",
                  "ms_per_tick": el / seq_ticks * 1e3,
                  "what": "the closed loop: one tick at a time on one context, its ordered heard links left in HBM "
                          "(rm_tick_run_sources_device; one launch per tick for the geometric media, rm_tick.hip)"}
    if run.env.rank == 0 and run.env.world == 1 and not run.as_rank:
        # the one-launch tick against the same roofline: the tick's required bytes over the kernel's own interval (the tick is a
        # chain of dependent round trips on a mostly idle device: bound by latency, neither by HBM nor by issue slots)
        try:
            cand1, _ = run.eng.slot_stats(0)
        except Exception:
            cand1 = 0
        rl = roofline_object(kernels=seq_kernels, n_samples=seq_samples, n_loc=run.n_loc, t_per_tick=run.t_per_tick, heard=heard, cand=cand1,
                             ticks_per_launch=1, step_s=sequential["ms_per_tick"] * 1e-3, contexts=1, workload=args.workload,
                             pmc_ok=(args.nodes == 0), tick_key=True, alone=seq_alone, sinr_column=run.sinr_column)
        rl["bound_note"] = ("latency: a lone tick of this size is a chain of dependent launches and memory round trips on a mostly idle "
                            "device; neither HBM bytes nor issue slots bind it (DESIGN.md section 4.7)")
        sequential["roofline"] = rl
    return sequential


def measure(args, env):
    """ONE configuration through the whole bench -- what the driver runs; rank 0 gets the result line's dictionary"""
    args = copy.copy(args)
    W, torch, dist = env.W, env.torch, env.dist
    run = setup(args, env)
    issue = pick_mode(run)
    elapsed, n_samples, kernels = timed_region(run, issue)

    eng, batch, world = run.eng, run.batch, env.world
    heard, dropped = run.last_run[0].batch_result_count(run.last_run[1]) if batch > 1 else eng.result_count()
    if dropped:
        raise SystemExit("heard links were dropped for capacity: the measurement is invalid")
    alone = alone_pass(run)
    tile_reuse = eng.batch_tile_reuse() if batch > 1 else 1
    if world > 1:   # the contract: MAX over ranks of the timed region
        rdev = env.dev if env.backend == "nccl" else torch.device("cpu")
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        hsum = torch.tensor([heard], dtype=torch.float64, device=rdev)
        dist.all_reduce(hsum, op=dist.ReduceOp.SUM)
        heard_total = float(hsum.item())
    else:
        heard_total = float(heard)
    links_per_tick = run.t_per_tick * (run.n - 1) if not run.as_rank else run.t_per_tick * run.n_loc
    timed_ticks = run.steps * batch
    value = links_per_tick * timed_ticks / elapsed
    if run.stateful and run.sharded is None and batch == 1:
        # the links the ticks resolved: the new frames against every receiver; the frames still on the air stay on the device
        # and are not swept again
        value = run.links_done / elapsed
    if run.stateful:
        inc, reb = eng.air_list_stats()
        ob, ot = eng.air_batch_stats()
        run.desc += (" -- %.2e link evaluations per tick (new frames only; the frames still on the air stay on the device: %d ticks in %d "
                     "batches found their interferers through the batch's index of the frames on the air, %d lone ticks among the frames by "
                     "scan, %d added their frames to per-receiver lists, %d rebuilt those)"
                     % (value * elapsed / timed_ticks, ot, ob, eng.air_scan_ticks(), inc, reb))
    sequential = None
    if (run.inflight > 1 or batch > 1) and run.sharded is None and world == 1:
        sequential = sequential_leg(run, links_per_tick, timed_ticks, heard)

    result = None
    if env.rank == 0:
        try:
            cand, _ = (run.last_run[0].slot_stats(run.last_run[1]) if batch > 1 else eng.slot_stats(0))
        except Exception:
            cand = 0
        step_s = elapsed / run.steps
        pairs = interferers = 0
        if run.stateful and batch > 1:
            pairs, _, interferers = eng.air_batch_pairs()
        roofline = roofline_object(kernels=kernels, n_samples=n_samples, n_loc=run.n_loc, t_per_tick=run.t_per_tick, heard=heard, cand=cand,
                                   ticks_per_launch=batch, step_s=step_s, contexts=run.inflight, workload=args.workload,
                                   pmc_ok=(world == 1 and not run.as_rank and args.nodes == 0), pairs_per_launch=pairs, alone=alone,
                                   sinr_column=run.sinr_column, tile_reuse=tile_reuse)
        if pairs:
            roofline["surviving_pairs_per_launch"] = pairs
            roofline["interfering_pairs_per_launch"] = interferers
        result = {
            "metric": baseline_metric(), "value": value, "unit": "links/s", "n_gpus": world, "steps": run.steps, "warmup": run.warmup,
            "ms_per_step": step_s * 1e3, "ms_per_tick": elapsed / timed_ticks * 1e3, "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": run.desc, "nodes": run.n, "tx_per_tick": run.t_per_tick, "tick_us": run.tick_us,
                       "ticks_in_flight": run.inflight * batch, "ticks_per_step": batch, "ticks_per_launch": batch, "contexts": run.inflight,
                       "step": "one launch sequence sweeping ticks_per_step simulated ticks",
                       "air_us": W.AIR_US, "medium": run.model, "heard_links_last_tick": heard_total, "candidate_links_last_tick": cand,
                       "sharding": ("receivers partitioned over %d ranks (%s), RCCL all-gather of Tx source indices per batch of ticks (%s)"
                                    % (world, "regions of the k-d order" if run.spatial else "node index ranges",
                                       "ncclAllGather inside libradiomedium_hip.so: stage + collective + frame list + sweep in one call"
                                       if run.lib_dist else "torch.distributed around the engine calls")) if world > 1 else "none"},
            "roofline": roofline,
        }
        if roofline["overlap_check"]["ok"] is False and not roofline["overlap_check"]["under_profiler"]:
            result["roofline_check_failed"] = ("the kernel intervals sampled inside the timed region add up to more than the driver-timed "
                                               "step allows (overlap_check): roofline.achieved / frac come from the launch sequences that ran alone")
        if sequential is not None:
            result["sequential_ticks"] = sequential
        # extra keys, each measured by its own function AFTER the timed region; none of them enters `value`
        if world == 1 and not run.stateful and not run.as_rank and not args.no_host_transfer:
            result["with_host_transfer"] = host_transfer_legs(env.rsa, W, eng, run.stream, torch, run.nodes, run.sources, run.n, run.t_per_tick,
                                                              run.tick_us, run.src_dev, run.pool, batch)
        if world == 1 and args.workload == "c3" and not args.no_scale_probe:
            result["at_1M_nodes"] = scale_probe(env.rsa, W, torch, env.dev, env.device_ordinal, run.inflight, batch)
            result["dense_layout"] = dense_probe(env.rsa, W, torch, env.dev, env.device_ordinal)
        if world == 1 and not args.no_cpu_baseline:
            st, mt = cpu_baseline(args.workload, run.nodes, run.sources, args.cpu_sample_ticks)
            result["cpu_baseline"] = st
            result["cpu_baseline_all_cores"] = mt
    for e in run.engines:
        e.close()
    return result


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.as_rank:
        spawn_ranks(args)
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner
    # on the first communicator), so everything but the result line goes to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    env = Env()
    env.rank = int(os.environ.get("RANK", "0"))
    env.world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("RM_BENCH_DRY_RUN") == "1":
        return dry_run(args, env.rank, env.world, result_fd)
    if args.inflight <= 0:
        args.inflight = 3
    args.batch_default = args.batch <= 0
    if args.batch <= 0:
        # several GPUs: a rank's share of a tick shrinks with the ranks, a batch's fixed costs (five launches, the collective)
        # do not: more ticks per launch sequence
        args.batch = 128 if env.world == 1 else min(512, 64 * env.world)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if env.world != args.gpus:
        args.gpus = env.world      # under a launcher the launcher's world size is the truth

    import torch
    import radio_sim_amd as rsa
    from radio_sim_amd import workload as W
    env.torch, env.rsa, env.W = torch, rsa, W
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # RM_FORCE_DEVICE / RM_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with one GPU
    env.device_ordinal = int(os.environ.get("RM_FORCE_DEVICE", local_rank))
    env.backend = os.environ.get("RM_DIST_BACKEND", "nccl")
    torch.cuda.set_device(env.device_ordinal)
    env.dev = torch.device("cuda", env.device_ordinal)
    env.dist = None
    # RM_DIST_SINGLE=1 with --force-sharded: a one-rank process group, so that the collectives of the
    # sharded driver go through RCCL itself on a box with one GPU
    if env.world > 1 or (args.force_sharded and os.environ.get("RM_DIST_SINGLE") == "1"):
        import torch.distributed as dist
        env.dist = dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:      # the launcher (torchrun) normally supplies it; a one-rank rehearsal picks a free one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if env.backend == "nccl":
            dist.init_process_group("nccl", rank=env.rank, world_size=env.world, device_id=env.dev)
        else:
            dist.init_process_group(env.backend, rank=env.rank, world_size=env.world)

    if args.dense_only:
        d = dense_probe(rsa, W, torch, env.dev, env.device_ordinal)
        lead = d["null_medium"]
        out = {"metric": baseline_metric(), "value": lead["value"], "unit": "links/s", "n_gpus": 1, "steps": 24, "warmup": 3,
               "ms_per_step": lead["ms_per_tick"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic", "config": {"workload": lead["workload"], "nodes": lead["nodes"], "tx_per_tick": lead["tx_per_tick"]},
               "roofline": lead["roofline"], "dense_layout": d}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        return

    out = measure(args, env)
    if env.world > 1 and args.scaling == "strong" and not args.no_weak_probe and not args.as_rank and args.nodes == 0:
        # the same run once more with the node count grown as sqrt(GPUs) (constant link evaluations per GPU and tick):
        # printed as an extra key, the headline stays the BASELINE config
        a2 = copy.copy(args)
        a2.scaling = "weak"
        a2.steps = 8 if args.steps <= 0 else min(args.steps, 8)
        a2.warmup = 2 if args.warmup < 0 else min(args.warmup, 2)
        a2.no_cpu_baseline = True
        try:
            w = measure(a2, env)
            if out is not None and w is not None:
                out["weak_scaling"] = {k: w[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "ms_per_tick", "config")}
        except Exception as e:   # the headline line is printed whatever happens to the extra pass
            if out is not None:
                out["weak_scaling"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if out is not None:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        sys.stdout.flush()
    if env.dist is not None:
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
