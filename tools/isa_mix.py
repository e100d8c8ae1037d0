"""Static instruction mix of one kernel from hipcc's -save-temps assembly (make -C radio-sim_amd/csrc asm, or
hipcc ... -save-temps=obj -c rm_filter.hip): opcodes by class, and -- the part that matters for a VALU-issue-bound sweep --
the instructions of the innermost loops (basic blocks that branch back to themselves or to an earlier label), where a
wave spends its time.  Issue cost per class from MI355X_MICROARCH.md (a wave64 VALU op takes 4 cycles on its SIMD; 32-bit
integer multiplies and 64-bit adds/shifts of the hash are quarter rate: 16; fp64 FMA/MUL half rate: 8).

    python tools/isa_mix.py /tmp/rm_filter-hip-amdgcn-amd-amdhsa-gfx950.s k_filter_wg_batch 'Li4ELb1' [out.json]
"""
import collections
import json
import re
import sys

QUARTER = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_i32")
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_sin_f32", "v_cos_f32")


def klass(op):
    if op.startswith("v_"):
        if op in QUARTER or op.startswith("v_mad_u64") or op.startswith("v_mad_i64"):
            return "valu_int_mul_quarter_rate", 16
        if op in TRANS:
            return "valu_transcendental", 16
        if "_f64" in op:
            return "valu_fp64", 8
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "valu_compare", 4
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane") or "permlane" in op or op.startswith("v_mbcnt"):
            return "valu_cross_lane", 4
        if "_f32" in op or "_f16" in op:
            return "valu_fp32", 4
        return "valu_int_logic", 4
    if op.startswith("ds_"):
        return "lds", 0
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem", 0
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait_sync", 0
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
        return "smem", 0
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch", 0
    if op.startswith("s_"):
        return "salu", 0
    return "other", 0


def main():
    path, name, variant = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else None
    lines = open(path).read().split("\n")
    # the kernel's body: from its label to .Lfunc_end / s_endpgm after it
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^_ZN2rm\w*%s\w*%s\w*:" % (re.escape(str(len(name)) + name), re.escape(variant)), ln):
            start = i
            label = ln.rstrip(":")
            break
    if start is None:
        raise SystemExit("kernel not found")
    body = []
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end"):
            break
        body.append(ln)
    # basic blocks by label
    blocks, cur, order = collections.OrderedDict(), ".entry", [".entry"]
    blocks[cur] = []
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        blocks[cur].append((op, t))
    index = {b: i for i, b in enumerate(order)}
    # loops: a block whose branch targets itself or an earlier block closes a loop over [target, block]
    in_loop = collections.Counter()
    loops = []
    for b, ins in blocks.items():
        for op, t in ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = t.split()[-1]
                if tgt in index and index[tgt] <= index[b]:
                    loops.append((tgt, b))
                    for k in order[index[tgt]:index[b] + 1]:
                        in_loop[k] += 1
    total, cyc_total = collections.Counter(), collections.Counter()
    for b, ins in blocks.items():
        for op, _ in ins:
            k, c = klass(op)
            total[k] += 1
            cyc_total[k] += c
    # leaf loops (no other loop inside): their bodies are what a wave executes over and over
    spans = sorted(set((index[t], index[b]) for t, b in loops))
    leaf = [sp for sp in spans if not any(o != sp and sp[0] <= o[0] and o[1] <= sp[1] for o in spans)]
    loop_rows = []
    for lo, hi in leaf:
        mix, cyc, ops = collections.Counter(), collections.Counter(), collections.Counter()
        for k in order[lo:hi + 1]:
            for op, _ in blocks[k]:
                kl, c = klass(op)
                mix[kl] += 1
                cyc[kl] += c
                ops[op] += 1
        loop_rows.append({"blocks": "%s .. %s" % (order[lo], order[hi]), "instructions": sum(mix.values()), "valu_issue_cycles": sum(cyc.values()),
                          "by_class": dict(mix), "valu_issue_cycles_by_class": {k: v for k, v in cyc.items() if v},
                          "top_opcodes": dict(ops.most_common(12))})
    loop_rows.sort(key=lambda r: -r["valu_issue_cycles"])
    res = {"kernel": label.split(":")[0], "source": path.split("/")[-1], "basic_blocks": len(blocks), "backward_branches": len(loops),
           "static_instructions": dict(total), "static_valu_issue_cycles_by_class": {k: v for k, v in cyc_total.items() if v},
           "leaf_loops_by_valu_issue_cycles_per_iteration": loop_rows[:8],
           "note": "static counts (one per instruction in the text), not execution counts; a leaf loop's numbers are per iteration of one "
                   "wave; cycles = issue cost on one SIMD (4 / 8 / 16 per wave instruction by class: MI355X_MICROARCH.md)"}
    s = json.dumps(res, indent=1)
    print(s)
    if out:
        open(out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
