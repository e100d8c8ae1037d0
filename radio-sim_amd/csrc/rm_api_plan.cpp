// rm_api_plan.cpp -- planning and launch sequence of ONE tick (any entry point), per-stage profiling.
#include "rm_host.hpp"

using namespace rmh;

namespace rmh {

int ensure_link_buffers(rm_context *c, TickSlot &ts, int feat)
{
    if (ts.alloc_cap == c->cap && ts.d_counters.p && (feat & ~ts.alloc_feat) == 0) return RM_OK;
    const size_t cap = size_t((c->cap + rm::kShards - 1) / rm::kShards) * rm::kShards;
    if (ts.alloc_cap != c->cap || !ts.d_counters.p) {
        RM_HIP(ts.d_counters.ensure(16));
        RM_HIP(hipMemsetAsync(ts.d_counters.p, 0, 16 * sizeof(uint32_t), c->stream));
        RM_HIP(ts.d_shards.ensure(2 * rm::kShards * rm::kShardStride));
        RM_HIP(hipMemsetAsync(ts.d_shards.p, 0, 2 * rm::kShards * rm::kShardStride * sizeof(uint32_t), c->stream));
        ts.parity = 0;
        RM_HIP(ts.d_st_pkt.ensure(cap));
        RM_HIP(ts.d_st_dst.ensure(cap));
        RM_HIP(ts.d_st_blk.ensure(cap));
        RM_HIP(ts.d_st_flags.ensure(cap));
        RM_HIP(ts.d_out_pkt.ensure(cap));
        RM_HIP(ts.d_out_dst.ensure(cap));
        RM_HIP(ts.d_out_verdict.ensure(cap));
        RM_HIP(ts.d_out_rssi.ensure(cap));
        RM_HIP(ts.d_a_pkt.ensure(cap));
        RM_HIP(ts.d_a_dst.ensure(cap));
        RM_HIP(ts.d_a_verdict.ensure(cap));
        RM_HIP(ts.d_a_rssi.ensure(cap));
        ts.alloc_feat = 0;
    }
    feat |= ts.alloc_feat;
    if (feat & kFeatPayload) {
        RM_HIP(ts.d_st_aux.ensure(cap));
        RM_HIP(ts.d_st_prob.ensure(cap));
        RM_HIP(ts.d_st_orig.ensure(cap));
    }
    if (feat & kFeatSinr) {
        RM_HIP(ts.d_st_next.ensure(cap));
        RM_HIP(ts.d_st_lin.ensure(cap));
        RM_HIP(ts.d_st_sinr.ensure(cap));
        RM_HIP(ts.d_st_coll.ensure(cap));
        RM_HIP(ts.d_a_sinr.ensure(cap));
        RM_HIP(ts.d_out_sinr.ensure(cap));
        RM_HIP(ts.d_a_e.ensure(cap));
    }
    if (feat & kFeatDraws) {
        RM_HIP(ts.d_out_prob.ensure(cap));
        RM_HIP(ts.d_a_prob.ensure(cap));
        RM_HIP(ts.d_draw_scan.ensure(cap + 1));
        RM_HIP(ts.d_scan_block.ensure(cap / 2048 + 2));
    }
    ts.alloc_cap = c->cap;
    ts.alloc_feat = feat;
    return RM_OK;
}

// Can the on-air lists on the device take this tick's new frames as they are?  `oldest` = AirLists::tick of the
// oldest frame still on the air (0: none).  Anything an old entry was computed from -- the receivers' positions,
// channels, the model -- shows up as a dirty receiver table or pre-filter.
uint32_t air_sub_cap(const rm_context *c) // entries per sub-ring: the link capacity over the sub-rings, as a power of two
{
    size_t sub = 64;
    while (sub * rm::kShards < size_t(c->cap)) sub <<= 1;
    return uint32_t(sub);
}

bool air_lists_current(const rm_context *c, int64_t t_begin, uint32_t oldest)
{
    static const bool off = [] {
        const char *e = std::getenv("RM_AIR_LISTS"); // 0: rebuild the lists from every frame on the air, every tick
        return e && std::atoi(e) == 0;
    }();
    const rm_context::AirLists &a = c->air;
    if (off || !a.valid || c->rx_dirty || c->prefilter_dirty || t_begin < a.last_t_begin) return false;
    if (a.tick + 1 >= rm::kAirTickMax) return false;
    if (oldest != 0 && (oldest > a.tick || a.tick + 1 - oldest >= rm::kAirTicks - 1)) return false;
    return a.pool.p != nullptr && a.sub_cap == air_sub_cap(c) && a.head.n >= size_t(std::max(c->n_rx, 1));
}

// frames per tick up to which a lone tick takes the one-frame-per-workgroup path (RM_FRAME_TICK=0: never)
static int frame_tick_max()
{
    static const int v = [] {
        const char *e = std::getenv("RM_FRAME_TICK");
        return e ? std::atoi(e) : 4096;
    }();
    return v;
}

// Can this tick of the SINR medium be evaluated by scan (rm_airscan.hip)?  Exactly when the one-launch tick applies to its
// new frames (rm::frame_tick_segment: sorted table with boxes, fp32 pre-filter, room for the frames' segments): then no
// per-receiver lists are needed at all.  RM_SINR_SCAN=0 / RM_SINR_FRAMES=0 (read per tick: tests switch them) keep the lists.
bool air_scan_applies(rm_context *c, int n_new)
{
    const char *e_scan = std::getenv("RM_SINR_SCAN"), *e_fr = std::getenv("RM_SINR_FRAMES");
    if ((e_scan && std::atoi(e_scan) == 0) || (e_fr && std::atoi(e_fr) == 0)) return false;
    if (!is_sinr(c) || n_new <= 0 || n_new > frame_tick_max() || c->n_rx <= 0) return false;
    if (!c->rx_sorted || c->f32_slack > 0.05) return false; // (LaunchCfg::sorted / bbox / f64_filter of prepare_tick)
    const int n_cnt = ((n_new + rm::kTxChunk - 1) / rm::kTxChunk) * rm::kTxChunk;
    if (n_cnt > rm::kFusedScanMax) return false;
    return (c->cap / 2u) / uint32_t(n_cnt) >= 64u;
}

int prepare_tick(rm_context *c, TickSlot &ts, TickPlan &plan, bool want_wg, const rm_tx_record *tx, int n_active,
                 int first_new, const int32_t *src_list, int64_t src_start_us, int64_t src_air_us, int air_mode,
                 uint32_t air_oldest, const rm::PlanKnobs *knobs_in)
{
    const rm::PlanKnobs knobs = knobs_in ? *knobs_in : rm::read_plan_knobs();
    const int n_new = n_active - first_new;
    RM_TRY(ev_flush_append(c)); // (the tick before, if its append was left for a drain that did not come: its records are about to go)
    ts.have_result = false;
    ts.compact_pending = false;
    ts.dense_pending = ts.dense_result = ts.dense_layout_pending = false;
    ts.last_n_new = n_new;
    RM_TRY(prepare_nodes(c));

    // (kAirBatch: the tick is swept as the medium without SINR; the SINR medium's buffers are there for the stages that follow)
    const bool sinr_medium = is_sinr(c);
    const bool sinr = sinr_medium && air_mode != kAirBatch;
    const bool stochastic = maybe_draws(c);
    RM_TRY(ensure_link_buffers(c, ts, ((!c->rx_sorted || sinr_medium) ? kFeatPayload : 0) | (sinr_medium ? kFeatSinr : 0) |
                                          (stochastic ? kFeatDraws : 0)));
    const int rx_count = c->n_rx;
    const bool partitioned = rx_count != c->n;
    ts.draws_pending = false;

    rm::TickDev &t = plan.t;
    t = rm::TickDev{};
    t.tx = tx;
    t.src_list = src_list;
    t.tx_build = src_list ? const_cast<rm_tx_record *>(tx) : nullptr;
    t.src_start_us = src_start_us;
    t.src_air_us = src_air_us;
    t.n_active = n_active;
    t.first_new = first_new;
    // caller records in device memory are not inspected by the host (rm_tick_run_device, rm_batch_run_device): without
    // the draw kernels a fractional txProbability in one of them cannot be honoured -- the kernels flag it
    if (c->host_src && !src_list) { // rm_tick_flush*: the records are read where the host put them (no copy engine in the chain)
        t.gather_src = c->host_src;
        t.gather_slots = std::max(n_active, 1);
        t.gather_stride = 0;
        t.tx_build = const_cast<rm_tx_record *>(tx);
    }
    t.check_txprob = (!stochastic && !src_list && c->dev_records_from_caller && c->params.kind != RM_MODEL_NULL && c->params.kind != RM_MODEL_UDGM_CONST) ? 1 : 0;
    t.first_eval = (sinr && air_mode != kAirScan) ? 0 : first_new;
    const bool nothing_to_sweep = (n_new <= 0 || rx_count <= 0); // no launch at all: the lists stay as they are
    if (sinr && air_mode == kAirRebuild && nothing_to_sweep) c->air.valid = false; // rebuilt with the next frames
    if (sinr && air_mode == kAirIncremental && nothing_to_sweep) c->air.last_t_begin = c->t_begin;
    if (sinr && air_mode == kAirScan) {
        // by scan: every frame on the air is in `tx`, nothing is kept per receiver -- and whatever the lists hold is stale now
        c->air.valid = false;
        if (!nothing_to_sweep) {
            c->air.scans++;
            rm::ScanDev &sd = plan.scan;
            sd = rm::ScanDev{};
            t.air_scan = 1;
            t.air.t_begin = c->t_begin;
            sd.level = model_dev(c).ld_level;
            const size_t frames = size_t(std::max(n_active, 1));
            const size_t cnt_len = size_t(rm::kSgCells) + 1 + rm::kSgMax;
            RM_HIP(ts.d_scan_xyzr.ensure(frames));
            RM_HIP(ts.d_scan_ch.ensure(frames));
            RM_HIP(ts.d_sg_every.ensure(frames));
            RM_HIP(ts.d_self_next.ensure(frames));
            if (!ts.d_sg_cnt.p) {
                RM_HIP(ts.d_sg_cnt.ensure(2 * cnt_len));
                RM_HIP(ts.d_sg_bxyzr.ensure(size_t(rm::kSgCells) * rm::kSgK));
                RM_HIP(ts.d_sg_bci.ensure(size_t(rm::kSgCells) * rm::kSgK));
                ts.sg_clean[0] = ts.sg_clean[1] = false;
            }
            if (ts.d_self_slot.n < size_t(std::max(c->n, 1))) {
                RM_HIP(ts.d_self_slot.ensure(size_t(std::max(c->n, 1))));
                RM_HIP(hipMemsetAsync(ts.d_self_slot.p, 0, ts.d_self_slot.n * sizeof(unsigned long long), c->stream));
            }
            if (++c->air.stamp == 0u) { // (the stamps have gone round: forget the old ones)
                c->air.stamp = 1u;
                RM_HIP(hipMemsetAsync(ts.d_self_slot.p, 0, ts.d_self_slot.n * sizeof(unsigned long long), c->stream));
            }
            const int par = ts.sg_parity;
            // this tick's counters were zeroed by the tick by scan before it (k_sinr_scan); anything else in between: by a fill
            if (!ts.sg_clean[par]) RM_HIP(hipMemsetAsync(ts.d_sg_cnt.p + size_t(par) * cnt_len, 0, cnt_len * sizeof(uint32_t), c->stream));
            ts.sg_clean[0] = ts.sg_clean[1] = false; // (this one is used now; the other is zeroed by this tick's second launch: launch_tick)
            ts.sg_parity = par ^ 1;
            sd.xyzr = ts.d_scan_xyzr.p;
            sd.ch = ts.d_scan_ch.p;
            sd.cnt = ts.d_sg_cnt.p + size_t(par) * cnt_len;
            sd.cnt_next = ts.d_sg_cnt.p + size_t(par ^ 1) * cnt_len;
            sd.bucket_xyzr = ts.d_sg_bxyzr.p;
            sd.bucket_ci = ts.d_sg_bci.p;
            sd.every = ts.d_sg_every.p;
            sd.self_slot = ts.d_self_slot.p;
            sd.self_next = ts.d_self_next.p;
            sd.stamp = c->air.stamp;
            sd.half = std::max(float(c->coord_bound), 1e-20f);
            sd.inv = float(rm::kSgG) / (2.0f * sd.half);
        }
    } else if (sinr && air_mode != kAirNone && !nothing_to_sweep) {
        // the lists that live across ticks: kAirIncremental -- `tx` holds the new frames only (first_new == 0);
        // kAirRebuild -- `tx` holds every frame on the air and all of them leave their entries again
        rm_context::AirLists &a = c->air;
        const size_t sub = air_sub_cap(c);
        if (a.sub_cap != sub || a.head.n < size_t(std::max(rx_count, 1)) || !a.pool.p) {
            if (air_mode == kAirIncremental) return fail(RM_ERR_STATE, "internal: on-air lists not allocated");
            RM_HIP(a.pool.ensure(sub * rm::kShards));
            RM_HIP(a.head.ensure(std::max(rx_count, 1)));
            RM_HIP(a.tail.ensure(size_t(rm::kShards) * rm::kShardStride));
            RM_HIP(a.mark.ensure(size_t(rm::kAirTicks) * rm::kShards));
            RM_HIP(a.bad.ensure(1));
            a.sub_cap = uint32_t(sub);
        }
        if (air_mode == kAirRebuild) {
            RM_HIP(hipMemsetAsync(a.head.p, 0, a.head.n * sizeof(unsigned long long), c->stream));
            RM_HIP(hipMemsetAsync(a.tail.p, 0, a.tail.n * sizeof(uint32_t), c->stream));
            RM_HIP(hipMemsetAsync(a.bad.p, 0, sizeof(uint32_t), c->stream));
            RM_HIP(hipMemsetAsync(a.mark.p + rm::kShards, 0, rm::kShards * sizeof(uint32_t), c->stream)); // tick 1 begins at 0
            a.tick = 0;
            a.rebuilds++;
        } else {
            a.incremental++;
        }
        a.tick++;
        a.valid = true;
        a.last_t_begin = c->t_begin;
        t.air.pool = a.pool.p;
        t.air.head = a.head.p;
        t.air.tail = a.tail.p;
        t.air.mark = a.mark.p;
        t.air.bad = a.bad.p;
        t.air.sub_mask = a.sub_cap - 1u;
        t.air.sub_shift = uint32_t(__builtin_ctz(a.sub_cap));
        t.air.tick = a.tick;
        t.air.wtick = (air_mode == kAirRebuild || air_oldest == 0) ? a.tick : air_oldest;
        t.air.t_begin = c->t_begin;
    }
    const int n_eval = n_active - t.first_eval;
    const int n_chunks = (n_eval + rm::kTxChunk - 1) / rm::kTxChunk;
    t.cnt_base = ((first_new - t.first_eval) / rm::kTxChunk) * rm::kTxChunk;
    t.shift = (first_new - t.first_eval) - t.cnt_base;
    t.n_cnt = n_chunks * rm::kTxChunk - t.cnt_base;
    t.n_rx = rx_count;
    const rm::ModelDev m = model_dev(c);
    rm::LaunchCfg &cfg = plan.cfg;
    cfg = rm::LaunchCfg{};
    cfg.stochastic = stochastic;
    cfg.f64_filter = c->f32_slack > 0.05 || (m.geo_cut > 0 && c->f32_slack > 0.05 * m.geo_cut);
    cfg.sorted = c->rx_sorted;
    cfg.bbox = c->rx_sorted && !cfg.f64_filter;
    cfg.shadow = c->shadow_tbl_valid && !cfg.f64_filter && !knobs.no_shadow_table;
    const int filter_mode = rm::plan_filter(t, cfg, want_wg, knobs); // fixes t.rpt / t.n_slabs

    const size_t cells = size_t(std::max(t.n_cnt, 0) / rm::kTxChunk) * std::max(t.n_slabs, 1) * 64;
    if (!c->rx_sorted) {
        RM_HIP(ts.d_cnt.ensure(std::max<size_t>(cells, 1)));
        RM_HIP(ts.d_off.ensure(std::max<size_t>(cells, 1)));
    }
    RM_HIP(ts.d_slot_tot.ensure(size_t(std::max(t.n_cnt, 0)) + 1));
    {
        // per-frame counters that kernels add to: zero-filled when (re)allocated, then kept zero by
        // k_filter (cursor: same tick; candidate totals: the other parity for the next tick)
        const size_t need = size_t(std::max(t.n_cnt, 0)) + 1;
        if (need > ts.d_cursor.n || 2 * need > ts.d_cand_tot.n) {
            RM_HIP(ts.d_cursor.ensure(need * 2));
            RM_HIP(ts.d_cand_tot.ensure(need * 4));
            RM_HIP(hipMemsetAsync(ts.d_cursor.p, 0, ts.d_cursor.n * 4, c->stream));
            RM_HIP(hipMemsetAsync(ts.d_cand_tot.p, 0, ts.d_cand_tot.n * 4, c->stream));
            ts.zero_len = 0;
        }
        RM_HIP(ts.d_seg_off.ensure(need + 1));
    }
    RM_HIP(ts.d_slot_off.ensure(size_t(std::max(t.n_cnt, 0)) + 2));
    RM_HIP(ts.d_pkt_interf.ensure(std::max(n_new, 1)));
    RM_HIP(ts.d_pkt_rng.ensure(std::max(n_new, 1)));
    RM_HIP(ts.d_pkt_draw_cnt.ensure(std::max(n_new, 1)));
    RM_HIP(ts.d_head.ensure(std::max(rx_count, 1)));
    if (!c->d_rng.p) {
        RM_HIP(c->d_rng.ensure(1));
        const uint64_t s0 = (uint64_t(0) ^ 0x5DEECE66Dull) & ((1ull << 48) - 1);
        RM_HIP(hipMemcpyAsync(c->d_rng.p, &s0, 8, hipMemcpyHostToDevice, c->stream));
        RM_HIP(hipStreamSynchronize(c->stream));
    }

    t.cnt = ts.d_cnt.p;
    t.off = ts.d_off.p;
    t.slot_tot = ts.d_slot_tot.p;
    t.slot_off = ts.d_slot_off.p;
    uint32_t *counters = ts.d_counters.p + 8 * ts.parity;
    t.stage_count = counters;
    t.next_counters = ts.d_counters.p + 8 * (ts.parity ^ 1);
    t.shard_count = ts.d_shards.p + size_t(ts.parity) * rm::kShards * rm::kShardStride;
    t.next_shard_count = ts.d_shards.p + size_t(ts.parity ^ 1) * rm::kShards * rm::kShardStride;
    t.cap = c->cap;
    t.shard_mask = (want_wg && filter_mode == rm::kFilterWg && t.rpt == 4) ? 63u : uint32_t(rm::kShards - 1);
    if (want_wg && filter_mode == rm::kFilterWg && partitioned) {
        // a receiver partition hands the exact stage proportionally fewer candidates per tick: fewer, fuller shards keep
        // its 256-entry chunks full (64 shards are tuned for ~50 k candidates of 100 k receivers: ~700 per shard)
        static const int fixed = [] {
            const char *e = std::getenv("RM_BATCH_SHARDS"); // developer knob: 8 / 16 / 32 / 64
            return e ? std::atoi(e) : 0;
        }();
        uint32_t shards = 64;
        while (shards > 8 && uint64_t(rx_count) * 64u < uint64_t(100000) * shards) shards >>= 1;
        if (fixed == 8 || fixed == 16 || fixed == 32 || fixed == 64) shards = uint32_t(fixed);
        t.shard_mask = shards - 1u;
    }
    t.seg_cap = uint32_t((size_t((c->cap + rm::kShards - 1) / rm::kShards) * rm::kShards) / (t.shard_mask + 1));
    t.use_matrix = cfg.sorted ? 0 : 1;
    t.cursor = ts.d_cursor.p;
    {
        const size_t half = ts.d_cand_tot.n / 2;
        t.cand_tot = ts.d_cand_tot.p + size_t(ts.parity) * half;
        t.cand_tot_next = ts.d_cand_tot.p + size_t(ts.parity ^ 1) * half;
    }
    t.seg_off = ts.d_seg_off.p;
    ts.zero_len = std::max(ts.zero_len, std::max(t.n_cnt, 0));
    t.zero_len = ts.zero_len;
    t.a_e = ts.d_a_e.p;
    if (filter_mode != rm::kFilterGrid) { // per-frame pre-filter records
        RM_HIP(ts.d_p_txf.ensure(std::max(n_eval, 1)));
        RM_HIP(ts.d_p_ch.ensure(std::max(n_eval, 1)));
        RM_HIP(ts.d_p_src.ensure(std::max(n_eval, 1)));
        RM_HIP(ts.d_p_inv.ensure(std::max(n_eval, 1)));
        t.p_txf = ts.d_p_txf.p;
        t.p_ch = ts.d_p_ch.p;
        t.p_src = ts.d_p_src.p;
        t.p_inv = ts.d_p_inv.p;
    }
    t.st_pkt = ts.d_st_pkt.p;
    t.st_dst = ts.d_st_dst.p;
    t.st_blk = ts.d_st_blk.p;
    t.st_aux = ts.d_st_aux.p;
    t.st_prob = ts.d_st_prob.p;
    t.st_orig = ts.d_st_orig.p;
    t.st_lin = ts.d_st_lin.p;
    t.st_sinr = ts.d_st_sinr.p;
    t.st_next = ts.d_st_next.p;
    t.st_flags = ts.d_st_flags.p;
    t.st_coll = ts.d_st_coll.p;
    t.head = ts.d_head.p;
    t.out_count = counters + 2;
    t.out_pkt = ts.d_out_pkt.p;
    t.out_dst = ts.d_out_dst.p;
    t.out_verdict = ts.d_out_verdict.p;
    t.out_rssi = ts.d_out_rssi.p;
    t.out_sinr = sinr_medium ? ts.d_out_sinr.p : nullptr; // only the SINR extension writes it: 8 of a record's 25 bytes
    t.out_prob = ts.d_out_prob.p;
    if (cfg.sorted) {
        t.a_pkt = ts.d_a_pkt.p;
        t.a_dst = ts.d_a_dst.p;
        t.a_verdict = ts.d_a_verdict.p;
        t.a_rssi = ts.d_a_rssi.p;
        t.a_sinr = ts.d_a_sinr.p;
        t.a_prob = ts.d_a_prob.p;
    } else { // engine order == node-index order: the scatter writes the final records directly
        t.a_pkt = t.out_pkt;
        t.a_dst = t.out_dst;
        t.a_verdict = t.out_verdict;
        t.a_rssi = t.out_rssi;
        t.a_sinr = t.out_sinr;
        t.a_prob = t.out_prob;
    }
    t.pkt_interference = ts.d_pkt_interf.p;
    t.draw_scan = ts.d_draw_scan.p;
    t.scan_block = ts.d_scan_block.p;
    t.rng_state = c->d_rng.p;
    t.pkt_rng = ts.d_pkt_rng.p;
    t.pkt_draw_cnt = ts.d_pkt_draw_cnt.p;

    ts.last = t;
    ts.last_links = 0;
    plan.sinr = sinr;
    plan.stochastic = stochastic;
    plan.partitioned = partitioned;
    plan.empty = (n_new <= 0 || rx_count <= 0);
    if (plan.empty) {
        // nothing to sweep: publish an empty result in this parity's counters
        RM_HIP(hipMemsetAsync(counters, 0, 8 * sizeof(uint32_t), c->stream));
        ts.have_result = true;
        return RM_OK;
    }
    ts.parity ^= 1; // the filter stage zeroes the other parity for the next tick
    // links resolved: every evaluated frame against every other node (T * (N-1)); for a receiver
    // partition the frame's own source may lie outside it, so the product is reported as is
    ts.last_links = (rx_count == c->n) ? int64_t(n_eval) * (rx_count - 1) : int64_t(n_eval) * rx_count;
    return RM_OK;
}

// the launch sequence of one prepared tick
int launch_tick(rm_context *c, TickSlot &ts, const TickPlan &plan)
{
    if (plan.empty) return RM_OK;
    const rm::TickDev &t = plan.t;
    const rm::LaunchCfg &cfg = plan.cfg;
    const bool sinr = plan.sinr, stochastic = plan.stochastic, partitioned = plan.partitioned;
    const int rx_count = t.n_rx;
    const rm::ModelDev m = model_dev(c);
    const rm::NodesDev nd = nodes_dev(c);
    hipStream_t s = c->stream;
    if (stochastic && partitioned && part_spatial(c)) RM_HIP(c->d_draw_nodes.ensure(size_t(c->cap) + 1));
    // The launch sequence.  On a sampled tick (rm_profile_enable) every kernel launch carries its own pair of events
    // (rm::KernelProbe); otherwise the stages are launched as ever (or, with RM_GRAPH=1, replayed from an instantiated
    // hipGraph keyed by the launch arguments).
    ProbeScope probe(c);
    rm_context::Sample *const smp = probe.smp;
    auto stage = [&](int id) -> int {
        sample_stage(smp, id);
        return RM_OK;
    };
    if (!sinr && rm::dense_tick_applies(t, cfg, m, nd, !part_spatial(c))) {
        // a medium in which a frame is heard by a large share of the nodes (the reference's default Null medium, a lossless
        // matrix, a range over most of the field): node-order evaluation and ordered compaction, rm_dense.hip
        const size_t cells = size_t(rm::dense_tick_cells(nd, t));
        RM_HIP(ts.d_cnt.ensure(std::max<size_t>(cells, 1)));
        RM_HIP(ts.d_off.ensure(std::max<size_t>(cells, 1)));
        RM_HIP(ts.d_dense_mask.ensure(std::max<size_t>(cells, 1) * 16));
        if (t.gather_src) RM_HIP(hipMemcpyAsync(t.tx_build, t.gather_src, size_t(t.n_active) * sizeof(rm_tx_record), hipMemcpyHostToDevice, s));
        RM_TRY(stage(RM_STAGE_FILTER));
        const char *e_lazy = std::getenv("RM_DENSE_LAZY"); // 0: the records at once, as before ABI version 5 (read per tick: tests)
        const bool lazy = !(e_lazy && std::atoi(e_lazy) == 0);
        RM_HIP(rm::launch_dense_tick(s, nd, m, t, ts.d_cnt.p, ts.d_off.p, ts.d_dense_mask.p, lazy));
        ts.dense_pending = lazy;
        ts.dense_layout_pending = lazy;
        ts.dense_result = true;
        ts.dense_rx_first = nd.rx_first;
        ts.dense_chunks = (nd.pos_span + 1023) / 1024;
        ts.compact_pending = false;
        ts.last.seg_ordered = 0;
        ts.last_model = m;
        ts.last_cfg = cfg;
        ts.have_result = true;
        return RM_OK;
    }
    const int seg_len = (t.n_active - t.first_new <= frame_tick_max()) ? rm::frame_tick_segment(t, cfg, m) : 0;
    if (t.air_scan && seg_len == 0) return fail(RM_ERR_STATE, "internal: a tick planned by scan cannot take the one-launch form");
    if (t.gather_src && seg_len == 0 && t.filter_mode != rm::kFilterWg) {
        // only the one-launch tick and the two-level filter's pre-pass read records from the host's block: the other
        // kernels want them in device memory first
        RM_HIP(hipMemcpyAsync(t.tx_build, t.gather_src, size_t(t.n_active) * sizeof(rm_tx_record), hipMemcpyHostToDevice, s));
    }
    auto sequence = [&]() -> int {
        const bool air = sinr && t.air.pool != nullptr;
        const bool air_in_prep = air && t.filter_mode == rm::kFilterWg; // k_tick_prep leaves the SELF entries and looks at the sticky flag
        if (air && !air_in_prep) RM_HIP(rm::launch_air_begin(s, t));
        else if (sinr && !air && !t.air_scan) RM_HIP(hipMemsetAsync(ts.d_head.p, 0xFF, size_t(rx_count) * sizeof(int32_t), s));
        if (seg_len > 0) {
            // the closed-loop tick: filter, exact evaluation and node order of a frame inside one workgroup
            // (rm_tick.hip) -- ONE launch; the compact arrays only for the draw kernels, or on demand
            RM_TRY(stage(RM_STAGE_FILTER));
            RM_HIP(rm::launch_tick_frames(s, nd, m, t, cfg, seg_len, t.air_scan ? &plan.scan : nullptr));
            if (stochastic) {
                RM_TRY(stage(RM_STAGE_REORDER));
                rm::TickDev tr = t;
                tr.seg_ordered = 1;
                RM_HIP(rm::launch_reorder(s, m, tr, cfg));
                RM_TRY(stage(RM_STAGE_DRAWS));
                RM_HIP(rm::launch_draws_scan(s, t));
                if (!partitioned) RM_HIP(rm::launch_draws_apply(s, m, t, nullptr, 1, 0));
                else if (part_spatial(c)) RM_HIP(rm::launch_draw_nodes(s, t, c->d_draw_nodes.p));
            }
            return RM_OK;
        }
        RM_TRY(stage(RM_STAGE_FILTER));
        if (rm::frames_cand_applies(t, cfg)) RM_HIP(rm::launch_frames_cand(s, nd, m, t, cfg)); // a frame finds its own receivers
        else RM_HIP(rm::launch_filter(s, nd, m, t, cfg));
        RM_TRY(stage(RM_STAGE_EXACT));
        RM_HIP(rm::launch_seg_scan(s, t));
        RM_HIP(rm::launch_exact(s, nd, m, t, cfg));
        if (sinr && !air_in_prep) {
            RM_TRY(stage(RM_STAGE_SELF));
            RM_HIP(rm::launch_self_entries(s, nd, t));
        }
        if (t.use_matrix || t.n_cnt > 8192) {
            RM_TRY(stage(RM_STAGE_OFFSETS));
            RM_HIP(rm::launch_offsets(s, t));
        }
        if (sinr && !(air && cfg.sorted)) { // sorted tables with the cross-tick lists: k_reorder walks the lists itself
            RM_TRY(stage(RM_STAGE_SINR));
            RM_HIP(rm::launch_sinr(s, m, t));
        }
        if (cfg.sorted) {
            RM_TRY(stage(RM_STAGE_REORDER));
            RM_HIP(rm::launch_reorder(s, m, t, cfg));
        } else {
            RM_TRY(stage(RM_STAGE_SCATTER));
            RM_HIP(rm::launch_finalize(s, nd, m, t, cfg));
        }
        if (stochastic) {
            RM_TRY(stage(RM_STAGE_DRAWS));
            RM_HIP(rm::launch_draws_scan(s, t));
            // a receiver partition sees only its share of every packet's draws: the caller exchanges
            // the per-packet counts (rm_draw_counts_device) and calls rm_tick_finish_draws
            if (!partitioned) RM_HIP(rm::launch_draws_apply(s, m, t, nullptr, 1, 0));
            else if (part_spatial(c)) RM_HIP(rm::launch_draw_nodes(s, t, c->d_draw_nodes.p)); // ... and the drawing links' nodes
        }
        return RM_OK;
    };
    if (c->use_graphs && !smp && !t.air_scan) { // (a tick by scan carries a new stamp every time: nothing to replay)
        uint64_t key = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t n) {
            const unsigned char *b = static_cast<const unsigned char *>(p);
            for (size_t i = 0; i < n; ++i) key = (key ^ b[i]) * 1099511628211ull;
        };
        mix(&nd, sizeof(nd));
        mix(&m, sizeof(m));
        mix(&t, sizeof(t));
        const int bits[8] = {cfg.f64_filter, cfg.stochastic, cfg.sorted, cfg.bbox, sinr, cfg.shadow, partitioned, seg_len};
        mix(bits, sizeof(bits));
        hipGraphExec_t exec = nullptr;
        for (auto &g : c->graphs)
            if (g.key == key) {
                exec = g.exec;
                g.last_use = ++c->graph_clock;
            }
        if (!exec) {
            RM_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const int rc = sequence();
            hipGraph_t graph = nullptr;
            const hipError_t e_end = hipStreamEndCapture(s, &graph);
            if (rc != RM_OK) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            if (e_end != hipSuccess) return fail(RM_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e_end));
            const hipError_t e_inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e_inst != hipSuccess) return fail(RM_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e_inst));
            if (c->graphs.size() >= 16) { // evict the least recently used
                size_t victim = 0;
                for (size_t i = 1; i < c->graphs.size(); ++i)
                    if (c->graphs[i].last_use < c->graphs[victim].last_use) victim = i;
                (void)hipGraphExecDestroy(c->graphs[victim].exec);
                c->graphs.erase(c->graphs.begin() + victim);
            }
            c->graphs.push_back({key, exec, ++c->graph_clock});
        }
        RM_HIP(hipGraphLaunch(exec, s));
    } else {
        RM_TRY(sequence());
    }

    if (stochastic && partitioned) {
        ts.draws_pending = true;
        ts.pending_model = m;
    }
    if (t.air_scan) ts.sg_clean[ts.sg_parity] = true; // (k_sinr_scan has zeroed the next tick's index counters)
    ts.compact_pending = seg_len > 0 && !stochastic;
    ts.last.seg_ordered = (seg_len > 0) ? 1 : 0;
    ts.last_model = m;
    ts.last_cfg = cfg;
    ts.have_result = true;
    return RM_OK;
}

// the compact packet-major arrays of a tick that so far only has its per-frame segments
// a dense tick that ended with its cells: their offsets, the packets' offsets, the totals
int dense_layout(rm_context *c, TickSlot &ts)
{
    if (!ts.dense_layout_pending) return RM_OK;
    RM_HIP(rm::launch_dense_layout(c->stream, ts.last_model, ts.last, ts.d_cnt.p, ts.d_off.p, ts.dense_chunks));
    ts.dense_layout_pending = false;
    return RM_OK;
}

int materialize(rm_context *c, TickSlot &ts)
{
    if (ts.dense_pending) { // the dense tick's records, from its cells' lane masks
        RM_TRY(dense_layout(c, ts));
        RM_HIP(rm::launch_dense_write(c->stream, ts.last_model, ts.last, ts.d_cnt.p, ts.d_off.p, ts.d_dense_mask.p, ts.dense_rx_first,
                                      ts.dense_chunks));
        ts.dense_pending = false;
    }
    if (!ts.compact_pending) return RM_OK;
    RM_HIP(rm::launch_reorder(c->stream, ts.last_model, ts.last, ts.last_cfg));
    ts.compact_pending = false;
    return RM_OK;
}

int run_tick(rm_context *c, const rm_tx_record *tx, int n_active, int first_new, const int32_t *src_list, int64_t src_start_us,
             int64_t src_air_us, int air_mode, uint32_t air_oldest, bool ev_may_wait)
{
    TickPlan plan;
    RM_TRY(prepare_tick(c, *c, plan, false, tx, n_active, first_new, src_list, src_start_us, src_air_us, air_mode, air_oldest));
    RM_TRY(launch_tick(c, *c, plan));
    if (c->ev.on && !c->draws_pending) {
        if (plan.empty && c->last_n_new > 0) {
            // a tick without receivers on this rank: its packets exist all the same (slot_off of an empty tick is not written)
            RM_HIP(hipMemsetAsync(c->d_slot_off.p, 0, (size_t(std::max(c->last.n_cnt, 0)) + 2) * sizeof(uint32_t), c->stream));
        }
        RM_TRY(ev_append(c, *c, ev_may_wait));
    }
    return RM_OK;
}

static bool probe_take(void *user, const char *kernel, hipEvent_t *start, hipEvent_t *stop)
{
    rm_context::Sample *sm = static_cast<rm_context::Sample *>(user);
    if (size_t(sm->n) == sm->k.size()) {
        rm_context::Sample::K nk;
        if (hipEventCreate(&nk.a) != hipSuccess) return false;
        if (hipEventCreate(&nk.b) != hipSuccess) {
            (void)hipEventDestroy(nk.a);
            return false;
        }
        sm->k.push_back(nk);
    }
    rm_context::Sample::K &k = sm->k[size_t(sm->n++)];
    k.stage = sm->cur_stage;
    k.name = kernel;
    *start = k.a;
    *stop = k.b;
    return true;
}

rm_context::Sample *begin_sample(rm_context *c)
{
    if (!c->profile || (c->tick_index++ % uint64_t(c->profile_every)) != 0) return nullptr;
    if (c->ev_used == c->ev_pool.size()) c->ev_pool.emplace_back();
    rm_context::Sample *sm = &c->ev_pool[c->ev_used++];
    sm->n = 0;
    sm->cur_stage = RM_STAGE_FILTER;
    rm::g_probe.take = probe_take;
    rm::g_probe.user = sm;
    return sm;
}

void end_sample()
{
    rm::g_probe.take = nullptr;
    rm::g_probe.user = nullptr;
}

// "(k_filter_wg_batch<4, true>)" as the launch site spells it -> "k_filter_wg_batch<4, true>" as rocprofv3 prints it
static std::string kernel_label(const char *site)
{
    std::string n(site ? site : "?");
    while (!n.empty() && (n.front() == '(' || n.front() == ' ')) n.erase(n.begin());
    while (!n.empty() && (n.back() == ')' || n.back() == ' ')) n.pop_back();
    return n;
}

int drain_profile(rm_context *c)
{
    for (size_t i = 0; i < c->ev_used; ++i) {
        rm_context::Sample &sm = c->ev_pool[i];
        for (int k = 0; k < sm.n; ++k) {
            const rm_context::Sample::K &pk = sm.k[size_t(k)];
            float ms = 0;
            RM_HIP(hipEventSynchronize(pk.b));
            RM_HIP(hipEventElapsedTime(&ms, pk.a, pk.b));
            if (pk.stage >= 0 && pk.stage < RM_PROFILE_STAGES) c->prof_ms[pk.stage] += ms;
            const std::string label = kernel_label(pk.name);
            rm_context::KernelTime *kt = nullptr;
            for (auto &e : c->prof_kernels)
                if (e.name == label) kt = &e;
            if (!kt) {
                c->prof_kernels.push_back({label, pk.stage, 0u, 0.0});
                kt = &c->prof_kernels.back();
            }
            kt->launches++;
            kt->ms += ms;
        }
        c->prof_samples++;
    }
    c->ev_used = 0;
    return RM_OK;
}

} // namespace rmh

extern "C" {

int rm_profile_enable(rm_context *c, int enable)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(drain_profile(c));
    c->profile = enable != 0;
    c->profile_every = enable > 0 ? enable : 1;
    c->tick_index = 0;
    c->prof_samples = 0;
    for (double &v : c->prof_ms) v = 0;
    c->prof_kernels.clear();
    return RM_OK;
}

int rm_profile_read(rm_context *c, uint32_t *samples, double *stage_ms)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(drain_profile(c));
    if (samples) *samples = c->prof_samples;
    if (stage_ms)
        for (int k = 0; k < RM_PROFILE_STAGES; ++k) stage_ms[k] = c->prof_ms[k];
    return RM_OK;
}

int rm_profile_kernels(rm_context *c, rm_kernel_time *out, int32_t cap, int32_t *count)
{
    if (!c || !count || cap < 0 || (cap > 0 && !out)) return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(drain_profile(c));
    *count = int32_t(c->prof_kernels.size());
    for (int32_t i = 0; i < cap && size_t(i) < c->prof_kernels.size(); ++i) {
        const rm_context::KernelTime &kt = c->prof_kernels[size_t(i)];
        std::memset(&out[i], 0, sizeof(out[i]));
        std::strncpy(out[i].name, kt.name.c_str(), sizeof(out[i].name) - 1);
        out[i].stage = kt.stage;
        out[i].launches = kt.launches;
        out[i].total_ms = kt.ms;
    }
    return RM_OK;
}

int64_t rm_last_link_evaluations(const rm_context *c) { return c ? c->last_links : 0; }

int rm_slot_stats(rm_context *c, int32_t slot, uint64_t *candidates, uint64_t *heard)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    TickSlot *ts = slot_of(c, slot);
    if (!ts || !ts->have_result) return fail(RM_ERR_STATE, "no evaluated tick in this result slot");
    RM_HIP(hipSetDevice(c->device));
    uint32_t count = 0, dropped = 0;
    RM_TRY(result_count(c, *ts, &count, &dropped));
    if (heard) *heard = count;
    if (candidates) {
        std::vector<uint32_t> sh(size_t(rm::kShards) * rm::kShardStride);
        RM_HIP(hipMemcpyAsync(sh.data(), ts->last.shard_count, sh.size() * 4, hipMemcpyDeviceToHost, c->stream));
        RM_HIP(hipStreamSynchronize(c->stream));
        uint64_t sum = 0;
        for (int k = 0; k < rm::kShards; ++k) sum += sh[size_t(k) * rm::kShardStride];
        *candidates = sum;
    }
    return RM_OK;
}

} // extern "C"
