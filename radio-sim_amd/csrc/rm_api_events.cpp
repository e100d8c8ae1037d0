// rm_api_events.cpp -- C ABI: the reception stage on the device (rm_events_*, rm_node_info).
#include "rm_host.hpp"

using namespace rmh;

namespace rmh {

rm::EvDev ev_dev(rm_context *c)
{
    rm::EvDev e{};
    rm_context::Events &v = c->ev;
    e.st = v.d_st.p;
    e.pk = v.d_pk.p;
    e.pk_mask = v.pk_cap - 1u;
    e.l_dst = v.d_ldst.p;
    e.l_rssi = v.d_lrssi.p;
    e.l_verdict = v.d_lverdict.p;
    e.pool_mask = v.pool_cap - 1u;
    e.g_time = v.d_gtime.p;
    e.g_meta = v.d_gmeta.p;
    e.g_ref = v.d_gref.p;
    e.g_rank = v.d_grank.p;
    e.cnt_by_rank = v.d_cnt.p;
    e.off_by_rank = v.d_off.p;
    e.g_run = v.d_grun.p;
    e.run_rec = v.d_run_rec.p;
    e.g_cap = v.g_cap;
    e.recv_key = v.d_recv_key.p;
    e.send_key = v.d_send_key.p;
    e.receiving = v.d_receiving.p;
    e.sending = v.d_sending.p;
    e.latched = v.d_latched.p;
    e.n_nodes = v.state_n;
    e.own_first = part_first(c);
    e.own_count = part_count(c);
    e.member = part_spatial(c) ? c->d_member.p : nullptr;
    e.par = v.par;
    return e;
}

// radio-state arrays for c->n nodes; what they hold for the nodes already known is kept
int ev_ensure_nodes(rm_context *c)
{
    rm_context::Events &v = c->ev;
    const int n = std::max(c->n, 1);
    if (v.state_n >= c->n && v.d_receiving.p) return RM_OK;
    const size_t old_n = size_t(v.state_n);
    RM_HIP(v.d_recv_key.ensure(size_t(n), true, c->stream));
    RM_HIP(v.d_send_key.ensure(size_t(n), true, c->stream));
    RM_HIP(v.d_receiving.ensure(size_t(n), true, c->stream));
    RM_HIP(v.d_sending.ensure(size_t(n), true, c->stream));
    RM_HIP(v.d_latched.ensure(size_t(n), true, c->stream));
    // DevBuf grows geometrically: clear everything behind the nodes that were there
    RM_HIP(hipMemsetAsync(v.d_recv_key.p + old_n, 0, (v.d_recv_key.n - old_n) * sizeof(unsigned long long), c->stream));
    RM_HIP(hipMemsetAsync(v.d_send_key.p + old_n, 0, (v.d_send_key.n - old_n) * sizeof(unsigned long long), c->stream));
    RM_HIP(hipMemsetAsync(v.d_receiving.p + old_n, 0, v.d_receiving.n - old_n, c->stream));
    RM_HIP(hipMemsetAsync(v.d_sending.p + old_n, 0, v.d_sending.n - old_n, c->stream));
    RM_HIP(hipMemsetAsync(v.d_latched.p + old_n, 0, (v.d_latched.n - old_n) * sizeof(double), c->stream));
    v.state_n = c->n;
    return RM_OK;
}

// hand the evaluated tick of slot `ts` to the reception stage (Simulator.generate*Events for every packet / heard link)
// the append that was left for the next drain, now on its own
int ev_flush_append(rm_context *c)
{
    rm_context::Events::Pending &p = c->ev.pending;
    if (!p.on) return RM_OK;
    p.on = false;
    RM_HIP(hipSetDevice(c->device));
    RM_HIP(rm::launch_ev_append(c->stream, ev_dev(c), p.ls, p.tx, p.n_new, p.now, p.immediate, p.dropped));
    c->ev.par ^= 1; // the launch wrote the other set of tails
    return RM_OK;
}

int ev_append(rm_context *c, TickSlot &ts, bool may_wait)
{
    if (!c->ev.on || !ts.have_result || ts.last_n_new <= 0) {
        return RM_OK;
    }
    RM_TRY(ev_flush_append(c));
    RM_TRY(ev_ensure_nodes(c));
    if (ts.dense_pending) RM_TRY(materialize(c, ts)); // (the reception stage reads records)
    const rm::TickDev &t = ts.last;
    rm::EvLinkSrc ls{};
    const uint32_t *dropped = nullptr;
    if (ts.compact_pending) {
        ls.dst = t.a_dst;
        ls.rssi = t.a_rssi;
        ls.verdict = t.a_verdict;
        ls.off = t.seg_off + t.shift;
        ls.cnt = t.cursor + t.shift;
        ls.n_scan = ts.last_n_new;
        dropped = t.stage_count + 1;
    } else {
        ls.dst = t.out_dst;
        ls.rssi = t.out_rssi;
        ls.verdict = t.out_verdict;
        ls.off = t.slot_off + t.shift;
        ls.cnt = nullptr;
        ls.n_scan = 0;
        dropped = t.out_count + 1;
    }
    ls.per_frame_verdict = (!ts.last_cfg.stochastic && !is_sinr(c)) ? 1 : 0;
    const int immediate = (c->params.kind == RM_MODEL_UDGM_CONST) ? 1 : 0;
    static const bool never_wait = [] { const char *e = std::getenv("RM_EV_FUSE"); return e && std::atoi(e) == 0; }();
    if (may_wait && !never_wait) { // (the closed loop: rm_events_process comes next, and takes the append into its first launch)
        rm_context::Events::Pending &p = c->ev.pending;
        p.on = true;
        p.ls = ls;
        p.tx = t.tx + t.first_new;
        p.n_new = ts.last_n_new;
        p.now = c->current_time;
        p.immediate = immediate;
        p.dropped = dropped;
        c->ev.next_packet += ts.last_n_new; // (numbered once the append is queued: a failed launch leaves the numbering where it was)
        return RM_OK;
    }
    RM_HIP(rm::launch_ev_append(c->stream, ev_dev(c), ls, t.tx + t.first_new, ts.last_n_new, c->current_time, immediate, dropped));
    c->ev.par ^= 1; // the launch wrote the other set of tails
    c->ev.next_packet += ts.last_n_new;
    return RM_OK;
}

} // namespace rmh

static size_t ev_out_bytes(uint32_t cap, uint32_t run_cap)
{
    return pad64(sizeof(rm::EvHeader)) + pad64(size_t(run_cap) * 8) + 2 * pad64(size_t(run_cap) * 4) + pad64(size_t(cap) * 4) + pad64(size_t(cap) * 8);
}

static rm::EvOut ev_out(rm_context *c)
{
    rm::EvOut o{};
    char *b = c->ev.h_out;
    const uint32_t cap = c->ev.pool_cap;
    size_t off = 0;
    const uint32_t run_cap = c->ev.g_cap; // (a run is a fired end group)
    o.hdr = reinterpret_cast<rm::EvHeader *>(b + off); off += pad64(sizeof(rm::EvHeader));
    o.run_packet = reinterpret_cast<int64_t *>(b + off); off += pad64(size_t(run_cap) * 8);
    o.run_first = reinterpret_cast<uint32_t *>(b + off); off += pad64(size_t(run_cap) * 4);
    o.run_count = reinterpret_cast<uint32_t *>(b + off); off += pad64(size_t(run_cap) * 4);
    o.dst = reinterpret_cast<int32_t *>(b + off); off += pad64(size_t(cap) * 4);
    o.rssi = reinterpret_cast<double *>(b + off);
    o.cap = cap;
    o.run_cap = run_cap;
    return o;
}

static uint32_t pow2_at_least(uint32_t v)
{
    uint32_t p = 64;
    while (p < v && p < (1u << 30)) p <<= 1;
    return p;
}

extern "C" {

int rm_events_disable(rm_context *c)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    rm_context::Events &v = c->ev;
    if (v.on || v.h_out) {
        (void)hipSetDevice(c->device);
        (void)ev_flush_append(c);
        (void)hipStreamSynchronize(c->stream);
    }
    v.pending.on = false;
    v.d_st.release(); v.d_pk.release(); v.d_ldst.release(); v.d_lrssi.release(); v.d_lverdict.release();
    v.d_gtime.release(); v.d_gmeta.release(); v.d_gref.release(); v.d_grank.release(); v.d_cnt.release(); v.d_off.release(); v.d_grun.release(); v.d_run_rec.release();
    v.d_recv_key.release(); v.d_send_key.release(); v.d_receiving.release(); v.d_sending.release(); v.d_latched.release();
    v.d_info_nodes.release();
    v.d_rep_rssi.release(); v.d_rep_sc.release(); v.d_rep_cnt.release();
    v.rep_n = 0;
    if (v.h_changed) (void)hipHostFree(v.h_changed);
    v.h_changed = nullptr;
    v.changed_n = 0;
    if (v.h_out) (void)hipHostFree(v.h_out);
    if (v.h_info) (void)hipHostFree(v.h_info);
    v.h_out = v.h_info = nullptr;
    v.info_n = 0;
    v.state_n = 0;
    v.on = false;
    v.next_packet = 0;
    return RM_OK;
}

int rm_events_enable(rm_context *c, uint32_t max_pending_packets, uint32_t max_pending_links)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    if (max_pending_links > (1u << 29)) return fail(RM_ERR_INVALID, "at most 2^29 pending links");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(rm_events_disable(c));
    rm_context::Events &v = c->ev;
    v.pk_cap = pow2_at_least(max_pending_packets ? max_pending_packets : (1u << 16));
    v.pool_cap = pow2_at_least(max_pending_links ? max_pending_links : (1u << 21));
    v.g_cap = 2u * v.pk_cap;
    RM_HIP(v.d_st.ensure(1));
    RM_HIP(v.d_pk.ensure(v.pk_cap));
    RM_HIP(v.d_ldst.ensure(v.pool_cap));
    RM_HIP(v.d_lrssi.ensure(v.pool_cap));
    RM_HIP(v.d_lverdict.ensure(v.pool_cap));
    RM_HIP(v.d_gtime.ensure(v.g_cap));
    RM_HIP(v.d_gmeta.ensure(v.g_cap));
    RM_HIP(v.d_gref.ensure(v.g_cap));
    RM_HIP(v.d_grank.ensure(v.g_cap));
    RM_HIP(v.d_cnt.ensure(v.g_cap));
    RM_HIP(v.d_off.ensure(v.g_cap));
    RM_HIP(v.d_grun.ensure(v.g_cap));
    RM_HIP(v.d_run_rec.ensure(2 * size_t(v.g_cap)));
    rm::EvState st{};
    st.top_max = int64_t(0x8000000000000000ull); // the top list is empty
    st.first_live = 0xFFFFFFFFu;
    RM_HIP(hipMemcpyAsync(v.d_st.p, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&v.h_out), ev_out_bytes(v.pool_cap, v.g_cap), hipHostMallocMapped));

    std::memset(v.h_out, 0, pad64(sizeof(rm::EvHeader)));
    v.seq = 0;
    v.on = true;
    v.next_packet = 0;
    v.oldest_packet = 0;
    v.par = 0;
    RM_TRY(ev_ensure_nodes(c));
    return RM_OK;
}

int64_t rm_events_next_packet(rm_context *c) { return (c && c->ev.on) ? c->ev.next_packet : -1; }

int rm_events_process(rm_context *c, int64_t time_us, rm_delivery_view *out)
{
    if (!c || !out) return fail(RM_ERR_INVALID, "NULL argument");
    if (!c->ev.on) return fail(RM_ERR_STATE, "rm_events_enable first");
    if (c->draws_pending) return fail(RM_ERR_STATE, "the last tick waits for rm_tick_finish_draws");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(ev_ensure_nodes(c));
    const rm::EvOut o = ev_out(c);
    const uint32_t seq = ++c->ev.seq;
    // (the ring window the drain looks at: at most the packets numbered since the oldest pending one of the last drain)
    const int64_t window = c->ev.next_packet - c->ev.oldest_packet;
    rm_context::Events::Pending &pend = c->ev.pending;
    if (pend.on) {
        pend.on = false;
        RM_HIP(rm::launch_ev_drain(c->stream, ev_dev(c), o, time_us, seq, uint32_t(std::min<int64_t>(std::max<int64_t>(window, 1), 0x7FFFFFFF)),
                                   &pend.ls, pend.tx, pend.n_new, pend.now, pend.immediate, pend.dropped));
        c->ev.par ^= 1; // (the first launch appended: it wrote the other set of tails)
    } else {
        RM_HIP(rm::launch_ev_drain(c->stream, ev_dev(c), o, time_us, seq, uint32_t(std::min<int64_t>(std::max<int64_t>(window, 1), 0x7FFFFFFF))));
    }
    c->current_time = time_us; // Simulator.java:156
    // (the header's four quarters carry the drain's number each, rm::EvHeader: all four have to have arrived)
    const uint32_t *const f0 = &o.hdr->seq, *const f1 = &o.hdr->seq1, *const f2 = &o.hdr->seq2, *const f3 = &o.hdr->seq3;
    auto arrived = [&]() {
        return __atomic_load_n(f0, __ATOMIC_ACQUIRE) == seq && __atomic_load_n(f1, __ATOMIC_ACQUIRE) == seq &&
               __atomic_load_n(f2, __ATOMIC_ACQUIRE) == seq && __atomic_load_n(f3, __ATOMIC_ACQUIRE) == seq;
    };
    bool seen = false;
    for (int spin = 0; spin < 400000 && !seen; ++spin) seen = arrived();
    if (!seen) {
        RM_HIP(hipStreamSynchronize(c->stream));
        if (!arrived()) return fail(RM_ERR_HIP, "the drain finished without its header");
    }
    out->count = o.hdr->count;
    out->pending_packets = o.hdr->pending_packets;
    out->oldest_packet = o.hdr->oldest_packet;
    c->ev.oldest_packet = o.hdr->oldest_packet;
    out->packet = nullptr; // (ABI version 3: the packet number comes once per run)
    out->dst = o.dst;
    out->rssi = o.rssi;
    out->n_runs = o.hdr->runs;
    out->run_packet = o.run_packet;
    out->run_first = o.run_first;
    out->run_count = o.run_count;
    if (o.hdr->err & 8u) c->air.valid = false;
    if (o.hdr->err & 8u) return fail(RM_ERR_CAPACITY, "a tick's heard links exceeded the link capacity (rm_set_link_capacity): its events are missing");
    if (o.hdr->err) return fail(RM_ERR_CAPACITY, "the reception stage ran out of room for pending packets / links (rm_events_enable)");
    if (o.hdr->total > o.hdr->count) return fail(RM_ERR_CAPACITY, "more deliveries than the delivery block holds");
    return RM_OK;
}

int rm_node_info(rm_context *c, const int32_t *nodes, int32_t n, double *rssi, int32_t *receiving, int32_t *channel)
{
    if (!c || n < 0) return fail(RM_ERR_INVALID, "bad arguments");
    if (!c->ev.on) return fail(RM_ERR_STATE, "rm_events_enable first");
    if (n == 0) return RM_OK;
    if (!nodes && n > c->n) return fail(RM_ERR_INVALID, "more nodes than the table holds");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(ev_flush_append(c)); // (an append left for the next drain runs with the state it was planned with, before anything reads or resets it)
    RM_TRY(ev_ensure_nodes(c));
    rm_context::Events &v = c->ev;
    if (v.info_n < n) {
        RM_HIP(hipStreamSynchronize(c->stream));
        if (v.h_info) RM_HIP(hipHostFree(v.h_info));
        v.h_info = nullptr;
        const int want = std::max(n + n / 2, 1024);
        RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&v.h_info), 64 + pad64(size_t(want) * 8) + 2 * pad64(size_t(want) * 4), hipHostMallocMapped));
        std::memset(v.h_info, 0, 64);
        v.info_n = want;
    }
    rm::NodeInfoOut o{};
    o.seq = reinterpret_cast<uint32_t *>(v.h_info);
    o.rssi = reinterpret_cast<double *>(v.h_info + 64);
    o.receiving = reinterpret_cast<int32_t *>(v.h_info + 64 + pad64(size_t(v.info_n) * 8));
    o.channel = reinterpret_cast<int32_t *>(v.h_info + 64 + pad64(size_t(v.info_n) * 8) + pad64(size_t(v.info_n) * 4));
    const int32_t *dev_nodes = nullptr;
    if (nodes) {
        RM_HIP(v.d_info_nodes.ensure(size_t(n)));
        RM_HIP(hipMemcpyAsync(v.d_info_nodes.p, nodes, size_t(n) * 4, hipMemcpyHostToDevice, c->stream));
        dev_nodes = v.d_info_nodes.p;
    }
    const uint32_t seq = ++v.info_seq;
    RM_HIP(rm::launch_node_info(c->stream, ev_dev(c), nodes_dev(c), dev_nodes, n, c->base_rssi, o, seq));
    volatile const uint32_t *flag = o.seq;
    bool seen = false;
    for (int spin = 0; spin < 400000 && !seen; ++spin) seen = (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq);
    if (!seen) RM_HIP(hipStreamSynchronize(c->stream));
    if (rssi) std::memcpy(rssi, o.rssi, size_t(n) * 8);
    if (receiving) std::memcpy(receiving, o.receiving, size_t(n) * 4);
    if (channel) std::memcpy(channel, o.channel, size_t(n) * 4);
    return RM_OK;
}

int rm_node_info_changed(rm_context *c, int32_t *nodes, double *rssi, int32_t *receiving, int32_t *channel, int32_t cap, int32_t *count)
{
    if (!c || !count || cap < 0 || (cap > 0 && (!nodes || !rssi || !receiving || !channel))) return fail(RM_ERR_INVALID, "bad arguments");
    if (!c->ev.on) return fail(RM_ERR_STATE, "rm_events_enable first");
    *count = 0;
    const int n = c->n;
    if (n == 0) return RM_OK;
    if (cap < n) return fail(RM_ERR_CAPACITY, "room for every node, please: a first call reports them all");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(ev_flush_append(c)); // (an append left for the next drain runs with the state it was planned with, before anything reads or resets it)
    RM_TRY(ev_ensure_nodes(c));
    rm_context::Events &v = c->ev;
    if (v.rep_n != n) { // a new table: nothing of it has been reported
        RM_HIP(v.d_rep_rssi.ensure(size_t(n)));
        RM_HIP(v.d_rep_sc.ensure(size_t(n)));
        RM_HIP(v.d_rep_cnt.ensure(2));
        RM_HIP(hipMemsetAsync(v.d_rep_sc.p, 0xFF, size_t(n) * sizeof(int2), c->stream)); // state -1
        RM_HIP(hipMemsetAsync(v.d_rep_cnt.p, 0, 2 * sizeof(uint32_t), c->stream));
        v.rep_n = n;
    }
    if (v.changed_n < n) {
        RM_HIP(hipStreamSynchronize(c->stream));
        if (v.h_changed) RM_HIP(hipHostFree(v.h_changed));
        v.h_changed = nullptr;
        const int want = std::max(n + n / 2, 1024);
        RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&v.h_changed), 64 + pad64(size_t(want) * 8) + 3 * pad64(size_t(want) * 4), hipHostMallocMapped));
        std::memset(v.h_changed, 0, 64);
        v.changed_n = want;
    }
    rm::NodeChangeOut o{};
    o.seq = reinterpret_cast<uint32_t *>(v.h_changed);
    o.count = o.seq + 1;
    o.rssi = reinterpret_cast<double *>(v.h_changed + 64);
    o.node = reinterpret_cast<int32_t *>(v.h_changed + 64 + pad64(size_t(v.changed_n) * 8));
    o.receiving = reinterpret_cast<int32_t *>(v.h_changed + 64 + pad64(size_t(v.changed_n) * 8) + pad64(size_t(v.changed_n) * 4));
    o.channel = reinterpret_cast<int32_t *>(v.h_changed + 64 + pad64(size_t(v.changed_n) * 8) + 2 * pad64(size_t(v.changed_n) * 4));
    const uint32_t seq = ++v.changed_seq;
    RM_HIP(rm::launch_node_info_changed(c->stream, ev_dev(c), nodes_dev(c), n, c->base_rssi, v.d_rep_rssi.p, v.d_rep_sc.p, o, uint32_t(v.changed_n),
                                        seq, v.d_rep_cnt.p));
    volatile const uint32_t *flag = o.seq;
    bool seen = false;
    for (int spin = 0; spin < 400000 && !seen; ++spin) seen = (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq);
    if (!seen) RM_HIP(hipStreamSynchronize(c->stream));
    const uint32_t k = std::min<uint32_t>(*o.count, uint32_t(n));
    std::memcpy(nodes, o.node, size_t(k) * 4);
    std::memcpy(rssi, o.rssi, size_t(k) * 8);
    std::memcpy(receiving, o.receiving, size_t(k) * 4);
    std::memcpy(channel, o.channel, size_t(k) * 4);
    *count = int32_t(k);
    return RM_OK;
}

} // extern "C"
