// rm_api_airbatch.cpp -- a batch of SINR ticks whose frames outlive their tick (BASELINE configs[4]) in ONE launch sequence:
// the heard links of all ticks from the batch sweep of the medium without SINR, then the interference stages of
// rm_airbatch.hip over the whole batch.  The frames stay where the lone ticks keep theirs: the on-air window of the context
// (d_air), so batches and lone ticks see each other's frames.
#include "rm_host.hpp"

using namespace rmh;

namespace {

struct AfterArg {
    rm::OvDev ov;
    rm::LaunchCfg cfg;
    int max_slot_frames, max_new, max_links;
};

// metres around the partition's boxes inside which a frame that stays on the air is kept for this rank (RM_RANK_MARGIN)
float rank_margin()
{
    const char *e = std::getenv("RM_RANK_MARGIN");
    const double v = e ? std::atof(e) : 64.0;
    return float(v > 0.0 ? v : 0.0);
}

int interference_stages(rm_context *c, void *arg)
{
    const AfterArg &a = *static_cast<const AfterArg *>(arg);
    const rm::ModelDev m = model_dev(c); // (the SINR medium itself: candidate level = min(sensitivity, interference floor))
    const rm::NodesDev nd = nodes_dev(c);
    RM_HIP(rm::launch_ov_index(c->stream, nd, m, a.ov, a.max_slot_frames));
    RM_HIP(rm::launch_ov_sinr(c->stream, nd, m, a.ov, a.max_new, a.max_links, a.cfg));
    return RM_OK;
}

} // namespace

namespace rmh {

// Does this batch of the SINR medium need the overlap form?  When a frame of an earlier call, or of an earlier tick of the
// batch, can still be on the air when a tick begins.  (Self-contained ticks keep the per-tick lists of rm_batch_*.)
bool overlap_wanted(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int32_t *n_per, const int64_t *start_us,
                    const int64_t *air_us)
{
    if (!is_sinr(c)) return false;
    for (const auto &bt : c->air_batches)
        if (bt.end_us > t_begin_us[0]) return true;
    for (int b = 0; b + 1 < n_ticks; ++b)
        if (n_per[b] > 0 && start_us[b] + air_us[b] > t_begin_us[b + 1]) return true;
    return false;
}

int batch_run_overlap(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us, const int32_t *const *dev_src,
                      const int32_t *n_per, const int64_t *start_us, const int64_t *air_us, const rm_tx_record *gathered, int gather_world,
                      int gather_slots, const int32_t *gathered_idx, int gather_block, int digest_off)
{
    if (gather_block <= 0) gather_block = n_ticks * gather_slots;
    // what the batched kernels carry: no java.util.Random draws (the draw stage walks compact records tick by tick and knows
    // nothing of verdicts that come later), time that does not run backwards inside the batch
    if (maybe_draws(c))
        return fail(RM_ERR_STATE, "the SINR medium carries frames that outlive their tick and its links can draw: run overlapping "
                                  "ticks one at a time");
    for (int b = 0; b < n_ticks; ++b) {
        if (b + 1 < n_ticks && t_begin_us[b + 1] < t_begin_us[b])
            return fail(RM_ERR_STATE, "overlapping SINR ticks of a batch have to be in time order");
        if (air_us[b] > int64_t(UINT32_MAX)) return fail(RM_ERR_INVALID, "a frame of the SINR medium has to be shorter than 2^32 us");
    }
    RM_TRY(prepare_nodes(c));
    if (!c->rx_sorted || c->n_rx <= 0)
        return fail(RM_ERR_STATE, "the SINR medium carries frames that outlive their tick into the next one: run overlapping ticks one "
                                  "at a time (the batched form needs the spatially sorted receiver table)");
    rm_context::Overlap &o = c->ov;
    RM_HIP(o.ticks.ensure(RM_MAX_BATCH));
    // a receiver partition keeps, of all ranks' gathered frames, those that can matter to its receivers (k_rank_frames) -- at the
    // interference floor, with a margin: these frames stay on the air while receivers may move (rm_api_batch.cpp)
    const bool rank_frames = gathered_idx != nullptr && rank_frames_wanted(c);
    // (refusals come before anything is planned or any window state moves: the ring of boxes the kept frames were selected against)
    if (rank_frames && c->air_culled && c->cull_end[c->cull_seq % uint32_t(rm::kCullRing)] > t_begin_us[0])
        return fail(RM_ERR_STATE, "more than 16 batches of frames selected for this partition are on the air at once: use larger batches");
    // the window: batches whose last frame has left the air go, and when the clock went back the frames that had already
    // left stay off (air_tick_device's rules)
    RM_TRY(air_window_expire(c, t_begin_us[0]));
    size_t total_new = 0;
    int max_new = 0;
    for (int b = 0; b < n_ticks; ++b) {
        total_new += size_t(n_per[b]);
        max_new = std::max(max_new, n_per[b]);
    }
    RM_TRY(air_window_reserve(c, total_new));
    const size_t live = c->air_tail - c->air_head;
    if (live > 0 && t_begin_us[0] < c->air_max_t_begin)
        RM_HIP(rm::launch_air_expire(c->stream, c->d_air.p + c->air_head, int(live), c->air_max_t_begin));
    // (air_max_t_begin is raised only once nothing can refuse the batch any more: a caller that is told to run the ticks one at
    // a time must find the window's clock where it was, or its first lone tick would retire frames that are still on the air
    // for the earlier ticks of the batch)
    c->air.valid = false; // (whatever the per-receiver lists hold is stale now)

    // ---- the ticks' plans: swept as the medium without SINR, every tick's frames built at its place of the window's tail
    while (c->extra_slots.size() + 1 < size_t(n_ticks)) c->extra_slots.emplace_back(new TickSlot());
    rm::PlanKnobs knobs = rm::read_plan_knobs();
    knobs.batch_ticks = n_ticks;
    static thread_local std::vector<TickSlot *> slots_v, run_slots;
    static thread_local std::vector<TickPlan> plans_v, run_plans;
    slots_v.resize(size_t(n_ticks));
    plans_v.resize(size_t(n_ticks));
    run_slots.clear();
    run_plans.clear();
    rm_tx_record *const base = c->d_air.p + c->air_head; // frame 0 of the index
    size_t at = live;
    rm::ModelDev ms = model_dev(c);
    ms.flags &= ~RM_LD_SINR;
    ms.ld_level = ms.ld_sens;
    bool batched = true;
    for (int b = 0; b < n_ticks; ++b) {
        TickSlot &ts = *slot_of(c, b);
        slots_v[size_t(b)] = &ts;
        TickPlan &pl = plans_v[size_t(b)];
        const rm_tx_record *tx = base + at;
        c->t_begin = t_begin_us[b];
        RM_TRY(prepare_tick(c, ts, pl, true, tx, n_per[b], 0, dev_src ? dev_src[b] : nullptr, dev_src ? start_us[b] : 0,
                            dev_src ? air_us[b] : 0, kAirBatch, 0, &knobs));
        if (gathered || gathered_idx) {
            rm::TickDev &t = pl.t;
            if (gathered_idx) {
                t.gather_idx = gathered_idx + size_t(b) * size_t(gather_slots);
                t.src_start_us = start_us[b];
                t.src_air_us = air_us[b];
            } else {
                t.gather_src = gathered + size_t(b) * size_t(gather_slots);
            }
            t.gather_slots = gather_slots;
            t.gather_stride = gather_block;
            t.tx_build = const_cast<rm_tx_record *>(tx);
            if (rank_frames && !pl.empty) {
                RM_TRY(plan_rank_frames(c, ts, t, n_per[b]));
                t.fl_pad = 1; // (the slots behind the listed frames: padding records, the window is walked as a whole)
                t.fl_ov_n_new = &o.ticks.p[b].n_new;
            }
            ts.last = t;
            if (t.n_pub > 0) ts.last.slot_off = ts.d_slot_off.p;
        }
        // (an empty tick has nothing to sweep and nothing to ask: it stays out of the launch, its result slot holds an empty result)
        if (!pl.empty) {
            batched = batched && rm::batch_eligible(pl.t, pl.cfg, ms) && (run_plans.empty() || pl.t.rpt == run_plans[0].t.rpt);
            run_slots.push_back(&ts);
            run_plans.push_back(pl);
        }
        at += size_t(n_per[b]);
    }
    if (run_plans.empty()) { // nothing transmits in the whole batch
        for (int b = 0; b < n_ticks; ++b) slots_v[size_t(b)]->have_result = true;
        c->air_max_t_begin = std::max(c->air_max_t_begin, t_begin_us[n_ticks - 1]);
        return RM_OK;
    }
    c->t_begin = t_begin_us[0];
    c->t_end = t_end_us[n_ticks - 1];
    if (!batched) {
        // (planned, never launched: a plan hands its slot's counters to the other parity because the tick's first kernel zeroes
        // them for the tick after it -- a tick that is not launched has zeroed nothing, so the slots go back to where they were, or
        // the next tick through the sweep kernels would start from the counters of the tick before last)
        for (int b = 0; b < n_ticks; ++b)
            if (!plans_v[size_t(b)].empty) slots_v[size_t(b)]->parity ^= 1;
        return fail(RM_ERR_STATE, "the SINR medium carries frames that outlive their tick into the next one: run overlapping ticks one "
                                  "at a time (the batched form takes non-empty ticks of at most 8192 frames over an fp32 frame)");
    }
    c->air_max_t_begin = std::max(c->air_max_t_begin, t_begin_us[n_ticks - 1]);

    // ---- the index's shape: time slots = the window's batches, then the ticks
    const int n_wslots = int(c->air_batches.size());
    const int n_slots = n_wslots + n_ticks;
    const size_t n_frames = live + total_new;
    const size_t n_bins = size_t(rm::kSgCells) * size_t(n_slots);
    const size_t desc_bytes = sizeof(rm::OvTick) * size_t(n_ticks) + sizeof(int32_t) * (size_t(n_slots) + 1);
    const int g = o.gen;
    o.gen ^= 1;
    if (o.h_desc_bytes[g] < desc_bytes) {
        if (o.h_desc[g]) {
            RM_HIP(hipEventSynchronize(o.h_ev[g]));
            RM_HIP(hipHostFree(o.h_desc[g]));
            o.h_desc[g] = nullptr;
            o.h_desc_bytes[g] = 0;
        }
        const size_t want = std::max<size_t>(desc_bytes * 2, 1 << 16);
        RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&o.h_desc[g]), want, hipHostMallocMapped)); // (read by the device itself: k_ov_begin)
        o.h_desc_bytes[g] = want;
        if (!o.h_ev[g]) RM_HIP(hipEventCreateWithFlags(&o.h_ev[g], hipEventDisableTiming));
    } else {
        RM_HIP(hipEventSynchronize(o.h_ev[g])); // (the copy that read this block last has completed)
    }
    if (!o.h_flag) {
        RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&o.h_flag), 128, hipHostMallocMapped)); // (written by the device itself: k_ov_verdict)
        o.h_flag[0] = o.h_flag[16] = 0u;
        for (int k = 0; k < 2; ++k) RM_HIP(hipEventCreateWithFlags(&o.h_flag_ev[k], hipEventDisableTiming));
    } else if (o.h_flag_used[g]) {
        RM_HIP(hipEventSynchronize(o.h_flag_ev[g])); // (the copy of the overflow flag of the batch that used this generation last has landed)
    }
    uint32_t *const h_flag = o.h_flag + 16 * g; // (one word per generation, a line apart: read here, written by that batch's last copy)
    rm::OvTick *const h_ticks = reinterpret_cast<rm::OvTick *>(o.h_desc[g]);
    int32_t *const h_first = reinterpret_cast<int32_t *>(o.h_desc[g] + sizeof(rm::OvTick) * size_t(n_ticks));
    std::vector<int64_t> slot_end;
    slot_end.resize(size_t(n_slots));
    int max_slot_frames = 1;
    {
        int f = 0;
        for (int s = 0; s < n_wslots; ++s) {
            h_first[s] = f;
            f += c->air_batches[size_t(s)].count;
            slot_end[size_t(s)] = c->air_batches[size_t(s)].end_us;
            max_slot_frames = std::max(max_slot_frames, c->air_batches[size_t(s)].count);
        }
        for (int b = 0; b < n_ticks; ++b) {
            h_first[n_wslots + b] = f;
            f += n_per[b];
            slot_end[size_t(n_wslots + b)] = start_us[b] + air_us[b];
        }
        h_first[n_slots] = f;
    }
    max_slot_frames = std::max(max_slot_frames, max_new);

    // ---- buffers of the index and of the pair list
    RM_HIP(o.fr_f.ensure(n_frames));
    RM_HIP(o.fr_m.ensure(n_frames));
    RM_HIP(o.fr_t.ensure(n_frames));
    RM_HIP(o.e_f.ensure(n_frames));
    RM_HIP(o.e_m.ensure(n_frames));
    RM_HIP(o.e_t.ensure(n_frames));
    RM_HIP(o.fr_bin.ensure(n_frames));
    RM_HIP(o.every.ensure(n_frames));
    RM_HIP(o.self_next.ensure(n_frames));
    RM_HIP(o.defer.ensure(n_frames));
    RM_HIP(o.items.ensure(std::max<size_t>(total_new, 1)));
    if (o.bin_cnt.n < n_bins) { // (the counts are zero between two batches: k_ov_fill takes every one back down)
        RM_HIP(o.bin_cnt.ensure(n_bins));
        RM_HIP(hipMemsetAsync(o.bin_cnt.p, 0, o.bin_cnt.n * sizeof(uint32_t), c->stream));
    }
    RM_HIP(o.bin_off.ensure(n_bins + 1));
    RM_HIP(o.block_sum.ensure((n_bins + rm::kOvScanBlock - 1) / rm::kOvScanBlock));
    RM_HIP(o.misc.ensure(8 + rm::kSgMax));
    RM_HIP(o.pair_tail.ensure(size_t(rm::kShards) * rm::kShardStride));
    RM_HIP(o.ticks.ensure(RM_MAX_BATCH));
    RM_HIP(o.slot_first.ensure(size_t(n_slots) + 1));
    if (c->d_self_slot.n < size_t(std::max(c->n, 1))) {
        RM_HIP(c->d_self_slot.ensure(size_t(std::max(c->n, 1))));
        RM_HIP(hipMemsetAsync(c->d_self_slot.p, 0, c->d_self_slot.n * sizeof(unsigned long long), c->stream));
    }
    if (++c->air.stamp == 0u) { // (the stamp of the nodes' chains is shared with the lone tick by scan; gone round: forget the old ones)
        c->air.stamp = 1u;
        RM_HIP(hipMemsetAsync(c->d_self_slot.p, 0, c->d_self_slot.n * sizeof(unsigned long long), c->stream));
    }
    {
        // the pair list: 200 - 450 surviving pairs per new frame at the bench's densities; doubled when the batch before ran out
        // (frames whose pairs do not fit are deferred to the in-place evaluation: slower, never wrong), up to 2 GB
        const char *e_cap = std::getenv("RM_OV_PAIR_CAP"); // (read per batch: tests set it)
        const size_t forced = e_cap ? size_t(std::atoll(e_cap)) : size_t(0);
        size_t want = o.pair_cap;
        if (want == 0) want = forced ? forced : std::max<size_t>(size_t(1) << 20, 512 * total_new); // (RM_OV_PAIR_CAP: tests start too small)
        else if (!forced) want = std::max(want, 512 * total_new);
        if (*h_flag != 0u) {
            want = std::max(want, std::min<size_t>(o.pair_cap * 2, size_t(1) << 27));
            *h_flag = 0u;
        }
        want = (want + rm::kShards - 1) / rm::kShards * rm::kShards;
        if (want > o.pair_cap) {
            RM_HIP(o.pairs.ensure(want));
            o.pair_cap = want;
        }
    }

    // ---- descriptors
    AfterArg arg{};
    rm::OvDev &ov = arg.ov;
    ov.tx = base;
    ov.n_frames = int(n_frames);
    ov.n_slots = n_slots;
    ov.n_ticks = n_ticks;
    ov.n_bins = int(n_bins);
    ov.max_new = std::max(max_new, 1);
    ov.slot_first = o.slot_first.p;
    ov.ticks = o.ticks.p;
    ov.fr_f = o.fr_f.p;
    ov.fr_m = o.fr_m.p;
    ov.fr_t = o.fr_t.p;
    ov.fr_bin = o.fr_bin.p;
    ov.bin_cnt = o.bin_cnt.p;
    ov.bin_off = o.bin_off.p;
    ov.block_sum = o.block_sum.p;
    ov.e_f = o.e_f.p;
    ov.e_m = o.e_m.p;
    ov.e_t = o.e_t.p;
    ov.every = o.every.p;
    ov.defer = o.defer.p;
    ov.misc = o.misc.p;
    ov.items = o.items.p;
    ov.self_slot = c->d_self_slot.p;
    ov.self_next = o.self_next.p;
    ov.stamp = c->air.stamp;
    ov.half = std::max(float(c->coord_bound), 1e-20f);
    ov.inv = float(rm::kSgG) / (2.0f * ov.half);
    ov.pairs = o.pairs.p;
    ov.pair_tail = o.pair_tail.p;
    ov.pair_seg = uint32_t(o.pair_cap / rm::kShards);
    ov.h_flag = h_flag;
    int slot_lo = 0;
    for (int b = 0; b < n_ticks; ++b) {
        const rm::TickDev &t = plans_v[size_t(b)].t;
        rm::OvTick &k = h_ticks[b];
        k = rm::OvTick{};
        k.frame_first = h_first[n_wslots + b];
        k.n_new = n_per[b];
        k.slot = n_wslots + b;
        // the oldest slot that may still hold a frame on the air when the tick begins (t_begin does not decrease over the
        // batch, so the bound only moves forward; the kernels test every frame's own times)
        while (slot_lo < k.slot && slot_end[size_t(slot_lo)] <= t_begin_us[b]) ++slot_lo;
        k.slot_lo = slot_lo;
        k.t_begin = t_begin_us[b];
        k.shift = t.shift;
        k.slot_off = t.slot_off;
        k.out_dst = t.out_dst;
        k.out_rssi = t.out_rssi;
        k.out_sinr = t.out_sinr;
        k.out_verdict = t.out_verdict;
        k.acc_lo = reinterpret_cast<unsigned long long *>(t.st_lin);
        k.acc_hi = reinterpret_cast<unsigned long long *>(t.st_sinr);
        k.hd = t.st_coll;
        k.flags = t.stage_count;
    }
    // the descriptors into device memory and the batch's counters to zero: one launch (the device reads the pinned block itself)
    RM_HIP(rm::launch_ov_begin(c->stream, h_ticks, h_first, n_ticks, n_slots, o.ticks.p, o.slot_first.p, o.misc.p, o.pair_tail.p));
    RM_HIP(hipEventRecord(o.h_ev[g], c->stream));
    arg.cfg = run_plans[0].cfg;
    arg.max_slot_frames = max_slot_frames;
    arg.max_new = max_new;
    arg.max_links = int(std::min<uint64_t>(c->cap, 1u << 30));

    rm::RankFramesArgs rf{};
    rf.gather_base = gathered_idx;
    rf.world = gather_world;
    rf.gather_block = gather_block;
    rf.digest_off = digest_off;
    rf.mine = c->table_digest;
    rf.margin = rank_margin();
    rf.use_chmask = 0; // (a receiver may change its channel while these frames are on the air)
    if (rank_frames) {
        // these frames stay on the air: the box they are selected against goes into the ring (rm::CullEntry), and the partition's
        // box as it is now is held against the entries of the batches that still have frames on the air
        if (!c->d_cull_ring.p || !c->air_culled) {
            RM_HIP(c->d_cull_ring.ensure(rm::kCullRing));
            std::vector<rm::CullEntry> none(rm::kCullRing);
            for (auto &e : none) e.end_us = INT64_MIN;
            RM_HIP(hipMemcpyAsync(c->d_cull_ring.p, none.data(), sizeof(rm::CullEntry) * rm::kCullRing, hipMemcpyHostToDevice, c->stream));
            RM_HIP(hipStreamSynchronize(c->stream)); // (the host vector goes away; once per window that was empty)
            for (auto &e : c->cull_end) e = INT64_MIN;
            c->air_culled = true;
        }
        const int slot = int(c->cull_seq % uint32_t(rm::kCullRing)); // (free: checked before anything was planned)
        int64_t batch_end = INT64_MIN;
        for (int b = 0; b < n_ticks; ++b)
            if (n_per[b] > 0) batch_end = std::max(batch_end, start_us[b] + air_us[b]);
        c->cull_end[slot] = batch_end;
        c->cull_seq++;
        rf.ring = c->d_cull_ring.p;
        rf.ring_slot = slot;
        rf.t_first = t_begin_us[0];
        rf.batch_end = batch_end;
    }
    const int rc = launch_batch(c, run_slots.data(), run_plans.data(), int(run_plans.size()), &ms, interference_stages, &arg,
                                (rank_frames || digest_off >= 0) && gathered_idx ? &rf : nullptr);
    if (rc != RM_OK) return rc;
    RM_HIP(hipEventRecord(o.h_flag_ev[g], c->stream)); // (k_ov_verdict has written this generation's overflow word)
    o.h_flag_used[g] = true;

    // ---- the batch's frames are on the air now
    c->air_tail += total_new;
    for (int b = 0; b < n_ticks; ++b)
        if (n_per[b] > 0) c->air_batches.push_back({n_per[b], start_us[b] + air_us[b], 0u}); // (0: not in the per-receiver lists)
    o.batches++;
    o.last_frames = n_frames;
    o.ticks_done += uint64_t(n_ticks);
    return RM_OK;
}

} // namespace rmh
