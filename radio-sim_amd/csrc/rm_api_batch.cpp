// rm_api_batch.cpp -- C ABI: several independent ticks per launch sequence (rm_batch_*), their results in one host-mapped block.
#include "rm_host.hpp"

#include <chrono>

using namespace rmh;

// RM_HOST_TIMING=1: what a batch call costs the host, phase by phase (stderr, every 16th call): the issue cost bounds a
// rank whose share of a tick needs less device time than the call takes (multi-GPU)
namespace {
struct HostClock {
    bool on = std::getenv("RM_HOST_TIMING") != nullptr;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    long calls = 0;
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void report()
    {
        if (!on || ++calls % 16) return;
        std::fprintf(stderr, "rm host timing per batch call [us]: checks %.1f  plan %.1f  descriptors %.1f (of which waiting for the batch "
                             "before last of this context: %.1f)  launches %.1f  total %.1f\n",
                     acc[0] / 16, acc[1] / 16, acc[2] / 16, acc[4] / 16, acc[3] / 16, (acc[0] + acc[1] + acc[2] + acc[3]) / 16);
        for (double &a : acc) a = 0;
    }
};
HostClock g_clock;
} // namespace

namespace rmh {

TickSlot *slot_of(rm_context *c, int32_t slot)
{
    if (slot == 0) return c;
    if (slot < 0 || size_t(slot) > c->extra_slots.size()) return nullptr;
    return c->extra_slots[size_t(slot) - 1].get();
}

// A batch of gathered source indices over a receiver partition keeps, per tick, only the frames that can matter to the
// partition's receivers (k_rank_frames).  RM_RANK_FRAMES=0: every rank keeps every frame, as before round 5 (read per call: tests).
bool rank_frames_wanted(rm_context *c)
{
    const char *e = std::getenv("RM_RANK_FRAMES");
    if (e && std::atoi(e) == 0) return false;
    return part_count(c) != c->n && part_count(c) > 0;
}

int plan_rank_frames(rm_context *c, TickSlot &ts, rm::TickDev &t, int n_pub)
{
    (void)c;
    RM_HIP(ts.d_fl_map.ensure(size_t(std::max(n_pub, 1))));
    RM_HIP(ts.d_fl_lb.ensure(size_t(std::max(n_pub, 1)) + 1));
    RM_HIP(ts.d_slot_off_loc.ensure(size_t(std::max(t.n_cnt, 0)) + 2));
    t.fl_map = ts.d_fl_map.p;
    t.fl_lb = ts.d_fl_lb.p;
    t.n_pub = n_pub;
    t.pub_off = ts.d_slot_off.p;       // what the result readers take for the packets' offsets: by global packet number
    t.slot_off = ts.d_slot_off_loc.p;  // the kernels' own: by listed frame
    return RM_OK;
}

} // namespace rmh

// the launch sequence of n prepared ticks in four launches (sorted table, fp32 frame)
int rmh::launch_batch(rm_context *c, TickSlot *const *slots, const TickPlan *plans, int n, const rm::ModelDev *m_override,
                      int (*after_sweep)(rm_context *, void *), void *after_arg, const rm::RankFramesArgs *rank_frames)
{
    const double tc0 = g_clock.on ? HostClock::now() : 0;
    static thread_local std::vector<rm::TickDev> ticks_v; // (RM_MAX_BATCH descriptors: not on the stack)
    ticks_v.resize(size_t(n));
    rm::TickDev *const ticks = ticks_v.data();
    for (int b = 0; b < n; ++b) ticks[b] = plans[b].t;
    {
        // large tables: the near-frame lists of the filter's phase A (k_near_lists), per tick and block of kNearSb workgroups.
        // RM_NEAR_LISTS=0 keeps phase A on all frames, 2 takes the lists for tables and ticks of any size (tests; read per batch).
        const char *e_knob = std::getenv("RM_NEAR_LISTS");
        const bool off = e_knob && std::atoi(e_knob) == 0, always = e_knob && std::atoi(e_knob) == 2;
        const rm::TickDev &t0 = ticks[0];
        const int n_wg = (t0.n_rx + rm::kGroup * 16 - 1) / (rm::kGroup * 16), n_sb = (n_wg + rm::kNearSb - 1) / rm::kNearSb;
        int max_eval = 0;
        bool same = true;
        for (int b = 0; b < n; ++b) {
            max_eval = std::max(max_eval, ticks[b].n_active - ticks[b].first_eval);
            same = same && ticks[b].filter_mode == rm::kFilterWg && ticks[b].rpt == 4 && ticks[b].n_rx == t0.n_rx;
        }
        const size_t entries = size_t(n) * size_t(n_sb) * size_t(max_eval);
        if (!off && same && plans[0].cfg.bbox && !plans[0].cfg.f64_filter && ((n_wg >= 4 * rm::kNearSb && max_eval >= 512) || (always && max_eval >= 1)) && entries <= (size_t(1) << 28)) {
            RM_HIP(c->d_near_list.ensure(entries));
            RM_HIP(c->d_near_cnt.ensure(size_t(n) * size_t(n_sb)));
            for (int b = 0; b < n; ++b) {
                ticks[b].near_list = c->d_near_list.p + size_t(b) * size_t(n_sb) * size_t(max_eval);
                ticks[b].near_cnt = c->d_near_cnt.p + size_t(b) * size_t(n_sb);
                ticks[b].near_cap = max_eval;
            }
        }
    }
    // the descriptors go to device memory (k_store_ticks, ordered on the stream after the previous
    // batch's kernels, which read the same array)
    RM_HIP(c->d_ticks.ensure(RM_MAX_BATCH));
    rm::TickDev *dev_ticks = c->d_ticks.p;
    const bool by_copy = n > 2 * 5; // beyond two k_store_ticks launches (five descriptors each): one fetch from pinned, host-mapped memory
    if (by_copy) {
        const int g = c->h_ticks_gen;
        c->h_ticks_gen ^= 1;
        if (!c->h_ticks[g]) {
            RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_ticks[g]), sizeof(rm::TickDev) * RM_MAX_BATCH, hipHostMallocMapped));
            RM_HIP(hipEventCreateWithFlags(&c->h_ticks_ev[g], hipEventDisableTiming));
        } else {
            const double tw0 = g_clock.on ? HostClock::now() : 0;
            RM_HIP(hipEventSynchronize(c->h_ticks_ev[g]));
            if (g_clock.on) g_clock.acc[4] += HostClock::now() - tw0;
        }
        std::memcpy(c->h_ticks[g], ticks, sizeof(rm::TickDev) * size_t(n));
        RM_HIP(rm::launch_fetch_ticks(c->stream, c->h_ticks[g], n, dev_ticks)); // the device reads the mapped block itself
        RM_HIP(hipEventRecord(c->h_ticks_ev[g], c->stream));
    }
    const rm::ModelDev m = m_override ? *m_override : model_dev(c);
    const rm::NodesDev nd = nodes_dev(c);
    const rm::LaunchCfg &cfg = plans[0].cfg;
    hipStream_t s = c->stream;
    const double tc1 = g_clock.on ? HostClock::now() : 0;
    if (g_clock.on) g_clock.acc[2] += tc1 - tc0;
    struct Done { // (every return below has issued what it will issue)
        double t0;
        ~Done() { if (g_clock.on) g_clock.acc[3] += HostClock::now() - t0; }
    } done{tc1};
    ProbeScope probe(c);
    rm_context::Sample *const smp = probe.smp;
    auto stage = [&](int id) -> int {
        sample_stage(smp, id);
        return RM_OK;
    };
    if (!by_copy) RM_HIP(rm::launch_store_ticks(s, ticks, n, dev_ticks));
    if (rank_frames) {
        // the ticks' frame lists: the descriptors' frame counts are the device's from here on (the medium itself, not the
        // sweep's override: the SINR medium's candidate level is its interference floor)
        RM_TRY(stage(RM_STAGE_FILTER));
        int max_frames = 1;
        for (int b = 0; b < n; ++b) max_frames = std::max(max_frames, ticks[b].n_pub);
        rm::RankFramesArgs rf = *rank_frames;
        rf.sweep_level = m.ld_level;
        RM_HIP(rm::launch_rank_frames(s, nd, model_dev(c), dev_ticks, n, max_frames, rf));
    }
    // RM_BATCH_FRAMES=1: the batch through the one-frame-per-workgroup kernel of the closed-loop tick instead of the
    // three sweep stages (one launch; the compact arrays on demand, per slot)
    static const bool batch_frames = std::getenv("RM_BATCH_FRAMES") != nullptr;
    if (batch_frames && !cfg.stochastic && !plans[0].sinr && !after_sweep) {
        int seg_len = rm::frame_tick_segment(ticks[0], cfg, m);
        for (int b = 1; b < n && seg_len > 0; ++b) seg_len = std::min(seg_len, rm::frame_tick_segment(ticks[b], cfg, m));
        if (seg_len > 0) {
            RM_TRY(stage(RM_STAGE_FILTER));
            RM_HIP(rm::launch_tick_frames_batch(s, nd, m, ticks, n, dev_ticks, cfg, seg_len));
            for (int b = 0; b < n; ++b) {
                slots[b]->have_result = true;
                slots[b]->compact_pending = true;
                slots[b]->last.seg_ordered = 1;
                slots[b]->last_model = m;
                slots[b]->last_cfg = cfg;
            }
            return RM_OK;
        }
    }
    RM_TRY(stage(RM_STAGE_FILTER));
    RM_HIP(rm::launch_batch_stage(s, 0, nd, m, ticks, n, dev_ticks, cfg));
    c->last_tile_reuse = rm::filter_ticks_per_wg(ticks[0], n);
    RM_TRY(stage(RM_STAGE_EXACT));
    RM_HIP(rm::launch_batch_stage(s, 1, nd, m, ticks, n, dev_ticks, cfg));
    if (plans[0].sinr && ticks[0].acc_lo == nullptr) { // (the per-receiver lists: SELF entries, then the walks; sums per receiver need neither)
        RM_TRY(stage(RM_STAGE_SINR));
        RM_HIP(rm::launch_batch_stage(s, 3, nd, m, ticks, n, dev_ticks, cfg));
    }
    if (plans[0].sinr && ticks[0].acc_lo != nullptr) { // sums per receiver: sinr and verdict of every heard link, full lanes
        RM_TRY(stage(RM_STAGE_SINR));
        RM_HIP(rm::launch_sinr_acc_batch(s, m, n, dev_ticks, int(std::min<uint32_t>(c->cap, 1u << 22)),
                                         (nd.n_rx > 0 && nd.n_rx < nd.n) ? std::max(1, nd.n / nd.n_rx) : 1));
    }
    RM_TRY(stage(RM_STAGE_REORDER));
    RM_HIP(rm::launch_batch_stage(s, 2, nd, m, ticks, n, dev_ticks, cfg));
    if (cfg.stochastic) {
        // the shared generator is walked tick by tick, in slot order, inside one launch
        RM_TRY(stage(RM_STAGE_DRAWS));
        RM_HIP(rm::launch_draws_batch(s, m, ticks, n, dev_ticks));
    }
    if (after_sweep) {
        RM_TRY(stage(RM_STAGE_SINR));
        RM_TRY(after_sweep(c, after_arg));
    }
    for (int b = 0; b < n; ++b) {
        slots[b]->have_result = true;
        slots[b]->compact_pending = false;
    }
    return RM_OK;
}

// gathered: the ticks' records as an all-gather of per-rank blocks [rank][tick][slot] left them (gather_world ranks,
// gather_slots records per rank and tick); every tick then has gather_world * gather_slots frames
int rmh::batch_run(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                   const int32_t *const *dev_src, const rm_tx_record *const *dev_new, const int32_t *n_per,
                   const int64_t *start_us, const int64_t *air_us, const rm_tx_record *gathered, int gather_world, int gather_slots,
                   const int32_t *gathered_idx, int gather_block, int digest_off)
{
    const bool any_gathered = gathered || gathered_idx;
    if (gather_block <= 0) gather_block = n_ticks * gather_slots; // (a rank's block of the gathered buffer, in its elements)
    if (!c || n_ticks < 1 || n_ticks > RM_MAX_BATCH || !t_begin_us || !t_end_us || (!n_per && !any_gathered) ||
        (!dev_src && !dev_new && !any_gathered) || ((dev_src || gathered_idx) && (!start_us || !air_us)) ||
        (any_gathered && (gather_world < 1 || gather_slots < 1)))
        return fail(RM_ERR_INVALID, "bad arguments");
    static thread_local std::vector<int32_t> n_gath;
    if (any_gathered) {
        n_gath.assign(size_t(n_ticks), gather_world * gather_slots);
        n_per = n_gath.data();
    }
    for (int b = 0; b < n_ticks; ++b)
        if (n_per[b] < 0 || (!any_gathered && n_per[b] > 0 && !(dev_src ? (const void *)dev_src[b] : (const void *)dev_new[b])) ||
            (dev_src && air_us[b] < 0))
            return fail(RM_ERR_INVALID, "bad arguments");
    const double th0 = g_clock.on ? HostClock::now() : 0;
    RM_HIP(hipSetDevice(c->device));
    const bool sinr = is_sinr(c);
    if (sinr) {
        // The SINR extension looks at every frame on the air.  Ticks whose frames are named by source indices carry their
        // time spans in the arguments: when a frame of an earlier call or of an earlier tick of the batch can still be on the
        // air the batch takes the overlap form (rm_api_airbatch.cpp: heard links by the sweep, interference over the whole
        // batch).  Self-contained ticks -- nothing of an earlier call and nothing of an earlier tick is still on the air when
        // a tick begins (e.g. air time <= tick length) -- keep their per-tick lists.  Records given by the caller are
        // verified on the device: every frame of tick b has to lie inside [t_begin[b], t_end[b]], and the ticks must not overlap.
        if (start_us && air_us && (dev_src || any_gathered) && overlap_wanted(c, n_ticks, t_begin_us, n_per, start_us, air_us))
            return batch_run_overlap(c, n_ticks, t_begin_us, t_end_us, dev_src, n_per, start_us, air_us, gathered, gather_world, gather_slots,
                                     gathered_idx, gather_block, digest_off);
        if (!dev_src && !gathered_idx)
            for (int b = 0; b < n_ticks; ++b)
                if (t_end_us[b] < t_begin_us[b] || (b + 1 < n_ticks && t_end_us[b] > t_begin_us[b + 1]))
                    return fail(RM_ERR_STATE, "SINR batches of records need ticks [t_begin, t_end] that do not overlap");
        for (const auto &bt : c->air_batches)
            if (bt.end_us > t_begin_us[0])
                return fail(RM_ERR_STATE, "frames of earlier calls are still on the air: run this tick on its own (or name the "
                                          "frames by source indices: rm_batch_run_sources_device takes overlapping ticks)");
        c->air.valid = false; // the ticks of a batch keep their lists to themselves
        c->air_batches.clear();
        c->air_head = c->air_tail = 0;
    }
    if (maybe_draws(c) && part_count(c) != c->n)
        return fail(RM_ERR_STATE, "a receiver partition whose links draw needs rm_tick_finish_draws per tick: run it one "
                                  "tick at a time");
    while (c->extra_slots.size() + 1 < size_t(n_ticks)) c->extra_slots.emplace_back(new TickSlot());
    rm::PlanKnobs knobs = rm::read_plan_knobs(); // once for the whole batch
    knobs.batch_ticks = n_ticks;
    const double th1 = g_clock.on ? HostClock::now() : 0;
    static thread_local std::vector<TickSlot *> slots_v;
    static thread_local std::vector<TickPlan> plans_v;
    slots_v.resize(size_t(n_ticks));
    plans_v.resize(size_t(n_ticks));
    TickSlot **const slots = slots_v.data();
    TickPlan *const plans = plans_v.data();
    bool batched = true;
    bool rank_frames = false;
    if (gathered_idx) {
        RM_TRY(prepare_nodes(c)); // (the partition's receivers: rank_frames_wanted asks for their number)
        rank_frames = rank_frames_wanted(c);
    }
    for (int b = 0; b < n_ticks; ++b) {
        TickSlot &ts = *slot_of(c, b);
        slots[b] = &ts;
        const rm_tx_record *tx = nullptr;
        if (dev_src && sinr && b == n_ticks - 1) {
            // the last tick's frames may outlive the batch: they are built where the on-air list of the
            // one-tick-at-a-time path lives
            RM_HIP(c->d_air.ensure(std::max<size_t>(size_t(n_per[b]), 1 << 16)));
            tx = c->d_air.p;
        } else if (dev_src || any_gathered) {
            RM_HIP(ts.d_tx.ensure(std::max(n_per[b], 1)));
            tx = ts.d_tx.p;
        } else {
            tx = dev_new[b];
        }
        c->dev_records_from_caller = (dev_src == nullptr && gathered_idx == nullptr);
        const int rc_prep = prepare_tick(c, ts, plans[b], true, tx, n_per[b], 0, dev_src ? dev_src[b] : nullptr,
                                         dev_src ? start_us[b] : 0, dev_src ? air_us[b] : 0, kAirNone, 0, &knobs);
        c->dev_records_from_caller = false;
        RM_TRY(rc_prep);
        if (any_gathered) {
            rm::TickDev &t = plans[b].t;
            if (gathered_idx) {
                t.gather_idx = gathered_idx + size_t(b) * size_t(gather_slots);
                t.src_start_us = start_us[b];
                t.src_air_us = air_us[b];
            } else {
                t.gather_src = gathered + size_t(b) * size_t(gather_slots);
            }
            t.gather_slots = gather_slots;
            t.gather_stride = gather_block;
            t.tx_build = ts.d_tx.p;
            if (rank_frames && !plans[b].empty) RM_TRY(plan_rank_frames(c, ts, t, n_per[b]));
            ts.last = t;
            if (t.n_pub > 0) ts.last.slot_off = ts.d_slot_off.p; // (result readers: the offsets by global packet number)
        }
        batched = batched && !plans[b].empty && rm::batch_eligible(plans[b].t, plans[b].cfg, model_dev(c)) &&
                  plans[b].t.rpt == plans[0].t.rpt;
    }
    c->t_begin = t_begin_us[0];
    c->t_end = t_end_us[n_ticks - 1];
    if (sinr && !dev_src && !gathered_idx)
        for (int b = 0; b < n_ticks; ++b) {
            plans[b].t.check_span = 1;
            plans[b].t.span_begin = t_begin_us[b];
            plans[b].t.span_end = t_end_us[b];
        }
    if (sinr && dev_src && n_per[n_ticks - 1] > 0) {
        c->air_tail = size_t(n_per[n_ticks - 1]);
        c->air_batches.push_back({n_per[n_ticks - 1], start_us[n_ticks - 1] + air_us[n_ticks - 1], 0u});
    }
    if (any_gathered && !batched) {
        for (int b = 0; b < n_ticks; ++b) // (planned, never launched: the slots' counter parity goes back, rm_api_airbatch.cpp)
            if (!plans[b].empty) slots[b]->parity ^= 1;
        return fail(RM_ERR_STATE, "gathered records go through the batched kernels only (sorted receiver table, fp32 frame, "
                                  "at most 8192 frames per tick, no empty tick)");
    }
    if (batched) {
        // SINR ticks named by source indices: one start and one air time per tick, so all of a tick's frames overlap each other and
        // a heard link's interference is its receiver's sum over ALL the tick's candidates less its own power -- summed per receiver
        // by the exact stage, no per-receiver lists (RM_SINR_ACC=0 keeps the lists; records given by the caller have their own
        // time spans and keep them too)
        const char *e_acc = std::getenv("RM_SINR_ACC");
        const bool acc = sinr && (dev_src || gathered_idx) && !(e_acc && std::atoi(e_acc) == 0);
        if (sinr)
            for (int b = 0; b < n_ticks; ++b) {
                rm::TickDev &t = plans[b].t;
                if (acc) {
                    t.acc_lo = reinterpret_cast<unsigned long long *>(t.st_lin);   // (link-sized buffers of the slot: room for every receiver)
                    t.acc_hi = reinterpret_cast<unsigned long long *>(t.st_sinr);
                    if (size_t(t.n_rx) > slots[b]->d_st_lin.n) {
                        for (int k = 0; k < n_ticks; ++k) // (planned, never launched: the slots' counter parity goes back)
                            if (!plans[k].empty) slots[k]->parity ^= 1;
                        return fail(RM_ERR_CAPACITY, "link capacity below the receiver count");
                    }
                } else {
                    t.reset_heads = 1;
                }
            }
        if (g_clock.on) {
            g_clock.acc[0] += th1 - th0;
            g_clock.acc[1] += HostClock::now() - th1;
        }
        rm::RankFramesArgs rf{};
        rf.gather_base = gathered_idx;
        rf.world = gather_world;
        rf.gather_block = gather_block;
        rf.digest_off = digest_off;
        rf.mine = c->table_digest;
        rf.margin = 0.f;     // nothing of these ticks outlives them
        rf.use_chmask = 1;
        const int rc = launch_batch(c, slots, plans, n_ticks, nullptr, nullptr, nullptr, (rank_frames || digest_off >= 0) && gathered_idx ? &rf : nullptr);
        g_clock.report();
        return rc;
    }
    // configurations the batched kernels do not cover (fp64 frame, unsorted table, very many frames,
    // empty ticks): the same ticks, one launch sequence each
    for (int b = 0; b < n_ticks; ++b) RM_TRY(launch_tick(c, *slots[b], plans[b]));
    return RM_OK;
}

extern "C" {

int rm_batch_run_sources_device(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                                const int32_t *const *dev_src, const int32_t *n_src, const int64_t *start_us,
                                const int64_t *air_us)
{
    if (!dev_src) return fail(RM_ERR_INVALID, "bad arguments");
    return batch_run(c, n_ticks, t_begin_us, t_end_us, dev_src, nullptr, n_src, start_us, air_us);
}

int rm_batch_run_device(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                        const rm_tx_record *const *dev_new, const int32_t *n_new)
{
    if (!dev_new) return fail(RM_ERR_INVALID, "bad arguments");
    return batch_run(c, n_ticks, t_begin_us, t_end_us, nullptr, dev_new, n_new, nullptr, nullptr);
}

int rm_batch_run_gathered_device(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                                 const rm_tx_record *dev_gathered, int32_t world, int32_t slots)
{
    if (!dev_gathered) return fail(RM_ERR_INVALID, "bad arguments");
    return batch_run(c, n_ticks, t_begin_us, t_end_us, nullptr, nullptr, nullptr, nullptr, nullptr, dev_gathered, world, slots);
}

int rm_batch_tile_reuse(const rm_context *c) { return c ? c->last_tile_reuse : fail(RM_ERR_INVALID, "ctx is NULL"); }

int rm_batch_result_device(rm_context *c, int32_t slot, rm_device_result *out)
{
    if (!c || !out) return fail(RM_ERR_INVALID, "NULL argument");
    TickSlot *ts = slot_of(c, slot);
    if (!ts) return fail(RM_ERR_INVALID, "no such result slot");
    return result_device(c, *ts, out);
}

int rm_batch_result_count(rm_context *c, int32_t slot, uint32_t *count, uint32_t *dropped)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    TickSlot *ts = slot_of(c, slot);
    if (!ts) return fail(RM_ERR_INVALID, "no such result slot");
    return result_count(c, *ts, count, dropped);
}

int rm_batch_result_copy(rm_context *c, int32_t slot, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi,
                         double *sinr, uint32_t cap, uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    TickSlot *ts = slot_of(c, slot);
    if (!ts) return fail(RM_ERR_INVALID, "no such result slot");
    if (!ts->have_result) return fail(RM_ERR_STATE, "no evaluated tick");
    RM_HIP(hipSetDevice(c->device));
    return copy_out(c, *ts, pkt, dst, verdict, rssi, sinr, cap, count, pkt_interference, pkt_offset);
}

int rm_batch_result_view(rm_context *c, int32_t n_slots, rm_host_result *out, int32_t *status)
{
    if (!c || !out || n_slots < 1 || n_slots > RM_MAX_BATCH) return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    if (!c->h_pack) RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_pack), sizeof(rm::PackSlot) * RM_MAX_BATCH, hipHostMallocDefault));
    RM_HIP(c->d_pack.ensure(RM_MAX_BATCH));
    uint32_t packets = 0;
    for (int b = 0; b < n_slots; ++b) {
        TickSlot *ts = slot_of(c, b);
        if (!ts || !ts->have_result) return fail(RM_ERR_STATE, "no evaluated tick in this result slot");
        if (ts->draws_pending) return fail(RM_ERR_STATE, "a slot's verdicts wait for rm_tick_finish_draws");
        RM_TRY(materialize(c, *ts));
        rm::PackSlot &ps = c->h_pack[b];
        ps.t = ts->last;
        ps.n_new = std::max(ts->last_n_new, 0);
        ps.have_offsets = (ps.n_new > 0 && part_count(c) > 0) ? 1 : 0;
        ps.pkt_base = packets;
        ps.pad = 0;
        packets += uint32_t(ps.n_new);
    }
    RM_TRY(ensure_stage(c, 0, packets + uint32_t(n_slots)));
    RM_HIP(hipMemcpyAsync(c->d_pack.p, c->h_pack, sizeof(rm::PackSlot) * size_t(n_slots), hipMemcpyHostToDevice, c->stream));
    rm::HostView v{};
    rm::BatchCounts *counts = nullptr;
    bool pkt_rssi = true; // one rssi per packet (the reference's media) if every slot allows it
    for (int b = 0; b < n_slots; ++b) pkt_rssi = pkt_rssi && host_pkt_rssi(c, *slot_of(c, b));
    for (int attempt = 0;; ++attempt) {
        v = stage_view(c->h_stage, c->stage_links, c->stage_packets, nullptr);
        if (pkt_rssi) v.rssi = nullptr;
        counts = stage_counts(c->h_stage);
        const uint32_t seq = ++c->stage_seq;
        RM_HIP(rm::launch_pack_batch(c->stream, c->d_pack.p, n_slots, v, counts, c->d_pack_done.p, seq));
        volatile const uint32_t *flag = &v.hdr->seq;
        bool seen = false;
        for (int spin = 0; spin < 400000 && !seen; ++spin) seen = (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq);
        if (!seen) RM_HIP(hipStreamSynchronize(c->stream));
        const uint64_t need = uint64_t(counts[n_slots - 1].link_base) + counts[n_slots - 1].stored;
        if (need <= v.links) break;
        if (attempt) return fail(RM_ERR_HIP, "result block could not be sized");
        RM_TRY(ensure_stage(c, uint32_t(std::min<uint64_t>(need + need / 4, 0xFFFFFFFFu)), packets + uint32_t(n_slots)));
    }
    int first_error = RM_OK;
    for (int b = 0; b < n_slots; ++b) {
        const rm::BatchCounts &bc = counts[b];
        const rm::PackSlot &ps = c->h_pack[b];
        rm_host_result &r = out[b];
        r.count = bc.stored;
        r.n_packets = uint32_t(ps.n_new);
        r.pkt_offset = v.pkt_offset + ps.pkt_base + uint32_t(b);
        r.pkt_interference = v.pkt_interference + ps.pkt_base;
        r.pkt = nullptr; // (ABI version 3: pkt_offset says it all)
        r.dst = v.dst + bc.link_base;
        r.verdict = v.verdict + bc.link_base;
        r.rssi = v.rssi ? v.rssi + bc.link_base : nullptr;
        r.pkt_rssi = v.rssi ? nullptr : v.pkt_rssi + ps.pkt_base;
        r.sinr = ps.t.out_sinr ? v.sinr + bc.link_base : nullptr;
        int st = RM_OK;
        if (bc.span_flag)
            st = fail(RM_ERR_STATE, record_flag_message(bc.span_flag));
        else if (bc.dropped)
            st = fail(RM_ERR_CAPACITY, "heard links exceed the context's link capacity (rm_set_link_capacity)");
        if (status) status[b] = st;
        if (st != RM_OK && first_error == RM_OK) first_error = st;
    }
    return first_error;
}

} // extern "C"
