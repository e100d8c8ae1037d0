// rm_api_tick.cpp -- C ABI: transmit(), the batched tick (host records and device-resident), results, java.util.Random draws across partitions.
#include "rm_host.hpp"

using namespace rmh;

namespace rmh {

rm_tx_record make_record(const rm_context *c, int32_t src, int64_t start_us, int64_t air_us, const double *txpower,
                         const int32_t *channel)
{
    rm_tx_record r;
    r.x = c->x[src];
    r.y = c->y[src];
    r.z = c->z[src];
    r.txpower = txpower ? *txpower : c->txpower[src];   // RadioPacket.java:49 / setTransmitPower
    r.txprob = c->txprob[src];
    r.start_us = start_us;
    r.air_us = air_us;
    r.src = src;
    r.channel = channel ? *channel : c->channel[src];   // RadioPacket.java:50 / setWirelessChannel
    return r;
}

// what the kernels' flag word (counters[6], HostHeader::span_flag) says about a tick's records
const char *record_flag_message(uint32_t flag)
{
    if (flag == 4u)
        return "a receiver of this partition has left the region (plus RM_RANK_MARGIN) that frames still on the air were selected "
               "for (it moved, or its radio was switched on): this rank no longer holds every frame that can interfere there -- let "
               "the frames leave the air, or run with RM_RANK_FRAMES=0";
    if (flag == 3u)
        return "the ranks' node tables differ (rm_table_digest): a rank built the frames' records from another copy of the node "
               "table than this context holds -- every rank needs every rm_node_update / rm_nodes_move of the others";
    return flag == 2u ? "a record given in device memory has a txprob strictly between 0 and 1, but no node probability asks for "
                        "java.util.Random draws: records in device memory must carry their source node's txprob"
                      : "a frame of this SINR tick lies outside the tick's [t_begin, t_end]: the batch was not self-contained";
}


int copy_out(rm_context *c, TickSlot &ts, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr,
                    uint32_t cap, uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset)
{
    if (ts.draws_pending)
        return fail(RM_ERR_STATE, "this rank's verdicts wait for the other ranks' draw counts: exchange "
                                  "rm_draw_counts_device and call rm_tick_finish_draws first");
    RM_TRY(materialize(c, ts));
    hipStream_t s = c->stream;
    uint32_t oc[5] = {0, 0, 0, 0, 0}; // [4]: a SINR tick of a batch held a frame outside its [t_begin, t_end]
    RM_HIP(hipMemcpyAsync(oc, ts.last.out_count, sizeof(oc), hipMemcpyDeviceToHost, s));
    RM_HIP(hipStreamSynchronize(s));
    if (oc[4]) return fail(RM_ERR_STATE, record_flag_message(oc[4]));
    if (count) *count = oc[2];
    const uint32_t k = std::min(oc[0], cap);
    if (k) {
        if (pkt) RM_HIP(hipMemcpyAsync(pkt, ts.d_out_pkt.p, k * 4ull, hipMemcpyDeviceToHost, s));
        if (dst) RM_HIP(hipMemcpyAsync(dst, ts.d_out_dst.p, k * 4ull, hipMemcpyDeviceToHost, s));
        if (verdict) RM_HIP(hipMemcpyAsync(verdict, ts.d_out_verdict.p, k, hipMemcpyDeviceToHost, s));
        if (rssi) RM_HIP(hipMemcpyAsync(rssi, ts.d_out_rssi.p, k * 8ull, hipMemcpyDeviceToHost, s));
        if (sinr && ts.last.out_sinr) RM_HIP(hipMemcpyAsync(sinr, ts.last.out_sinr, k * 8ull, hipMemcpyDeviceToHost, s));
        else if (sinr) std::memset(sinr, 0, k * 8ull);
    }
    const int n_new = ts.last_n_new;
    if (pkt_interference && n_new > 0)
        RM_HIP(hipMemcpyAsync(pkt_interference, ts.d_pkt_interf.p, size_t(n_new), hipMemcpyDeviceToHost, s));
    if (pkt_offset) {
        if (n_new > 0 && part_count(c) > 0)
            RM_HIP(hipMemcpyAsync(pkt_offset, ts.d_slot_off.p + ts.last.shift, (size_t(n_new) + 1) * 4,
                                  hipMemcpyDeviceToHost, s));
        else
            for (int i = 0; i <= std::max(n_new, 0); ++i) pkt_offset[i] = 0;
    }
    RM_HIP(hipStreamSynchronize(s));
    if (oc[1]) {
        c->air.valid = false; // a dropped SINR tick leaves the on-air lists incomplete
        return fail(RM_ERR_CAPACITY, "heard links exceed the context's link capacity (rm_set_link_capacity)");
    }
    if (oc[2] > cap) return fail(RM_ERR_CAPACITY, "caller buffers too small for the heard links");
    return RM_OK;
}

// ---- a flushed tick's result in host-mapped memory --------------------------------------------------

size_t pad64(size_t v) { return (v + 63) & ~size_t(63); }

rm::HostView stage_view(char *base, uint32_t links, uint32_t packets, size_t *bytes)
{
    rm::HostView v{};
    size_t o = 0;
    v.hdr = reinterpret_cast<rm::HostHeader *>(base + o); o += pad64(sizeof(rm::HostHeader));
    o += pad64(sizeof(rm::BatchCounts) * RM_MAX_BATCH); // per-slot counts of rm_batch_result_view (stage_counts)
    v.pkt_offset = reinterpret_cast<uint32_t *>(base + o); o += pad64((size_t(packets) + 1) * 4);
    v.pkt_interference = reinterpret_cast<uint8_t *>(base + o); o += pad64(size_t(packets) + 1);
    v.pkt_rssi = reinterpret_cast<double *>(base + o); o += pad64((size_t(packets) + 1) * 8);
    v.dst = reinterpret_cast<int32_t *>(base + o); o += pad64(size_t(links) * 4);
    v.rssi = reinterpret_cast<double *>(base + o); o += pad64(size_t(links) * 8);
    v.sinr = reinterpret_cast<double *>(base + o); o += pad64(size_t(links) * 8);
    v.verdict = reinterpret_cast<uint8_t *>(base + o); o += pad64(size_t(links) + 1);
    v.links = links;
    v.packets = packets;
    if (bytes) *bytes = o;
    return v;
}

rm::BatchCounts *stage_counts(char *base) { return reinterpret_cast<rm::BatchCounts *>(base + pad64(sizeof(rm::HostHeader))); }

int ensure_stage(rm_context *c, uint32_t links, uint32_t packets)
{
    if (c->h_stage && links <= c->stage_links && packets <= c->stage_packets) return RM_OK;
    links = std::max(links, std::max(c->stage_links, 1u << 16));
    packets = std::max(packets, std::max(c->stage_packets, 1u << 12));
    size_t bytes = 0;
    (void)stage_view(nullptr, links, packets, &bytes);
    RM_HIP(hipStreamSynchronize(c->stream)); // nothing may still write the old block
    if (c->h_stage) RM_HIP(hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->stage_links = c->stage_packets = 0;
    RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_stage), bytes, hipHostMallocMapped));
    std::memset(c->h_stage, 0, pad64(sizeof(rm::HostHeader)));
    c->stage_links = links;
    c->stage_packets = packets;
    if (!c->d_pack_done.p) {
        RM_HIP(c->d_pack_done.ensure(1));
        RM_HIP(hipMemsetAsync(c->d_pack_done.p, 0, sizeof(uint32_t), c->stream));
    }
    return RM_OK;
}

bool host_pkt_rssi(const rm_context *c, const TickSlot &ts)
{
    static const bool keep = std::getenv("RM_HOST_LINK_RSSI") != nullptr;
    // (a rank's frame list keeps the records of its listed frames only: an unlisted packet's power is not at hand)
    return !keep && c->params.kind != RM_MODEL_LOGDIST && ts.last.n_pub == 0;
}

// pack the evaluated tick of slot `ts` into the host-mapped block and wait for it
int pack_to_stage(rm_context *c, TickSlot &ts, rm::HostView *view)
{
    if (ts.draws_pending)
        return fail(RM_ERR_STATE, "this rank's verdicts wait for the other ranks' draw counts: exchange "
                                  "rm_draw_counts_device and call rm_tick_finish_draws first");
    const int n_new = std::max(ts.last_n_new, 0);
    const int have_offsets = (n_new > 0 && part_count(c) > 0) ? 1 : 0;
    if (ts.dense_pending) RM_TRY(materialize(c, ts));
    RM_TRY(ensure_stage(c, 0, uint32_t(n_new)));
    for (int attempt = 0; attempt < 2; ++attempt) {
        rm::HostView v = stage_view(c->h_stage, c->stage_links, c->stage_packets, nullptr);
        if (host_pkt_rssi(c, ts)) v.rssi = nullptr; // (the packets' transmit power instead: pkt_rssi)
        const uint32_t seq = ++c->stage_seq;
        if (ts.compact_pending) // straight from the frames' segments: no compact arrays in between
            RM_HIP(rm::launch_pack_frames(c->stream, ts.last_model, ts.last, n_new, v, c->d_pack_done.p, seq));
        else
            RM_HIP(rm::launch_pack_tick(c->stream, ts.last, n_new, have_offsets, v, c->d_pack_done.p, seq));
        // poll the sequence number (the kernel publishes it after everything else); a stream
        // synchronisation bounds the wait
        volatile const uint32_t *flag = &v.hdr->seq;
        bool seen = false;
        for (int spin = 0; spin < 400000 && !seen; ++spin) seen = (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq);
        if (!seen) RM_HIP(hipStreamSynchronize(c->stream));
        if (v.hdr->total <= v.links || v.hdr->dropped) {
            *view = v;
            return RM_OK;
        }
        RM_TRY(ensure_stage(c, std::max(v.hdr->total, 2u * v.links), uint32_t(n_new))); // more links than the block held
    }
    return fail(RM_ERR_HIP, "result block could not be sized");
}

int stage_status(rm_context *c, const rm::HostView &v)
{
    if (v.hdr->span_flag) return fail(RM_ERR_STATE, record_flag_message(v.hdr->span_flag));
    if (v.hdr->dropped) {
        c->air.valid = false;
        return fail(RM_ERR_CAPACITY, "heard links exceed the context's link capacity (rm_set_link_capacity)");
    }
    return RM_OK;
}

// evaluate the tick enqueued with rm_tick_begin / rm_enqueue_tx* (results stay on the device)
int tick_run_host(rm_context *c)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    if (!c->in_tick) return fail(RM_ERR_STATE, "rm_tick_flush without rm_tick_begin");
    RM_HIP(hipSetDevice(c->device));
    c->in_tick = false;
    // SINR: the frames of earlier ticks stay on the device (the window air_tick_device keeps, shared with the ticks whose
    // frames are in device memory already); only the new records cross the link
    const bool sinr = is_sinr(c);
    if (sinr)
        for (const rm_tx_record &r : c->pending)
            if (r.air_us < 0 || r.air_us > int64_t(UINT32_MAX))
                return fail(RM_ERR_INVALID, "a frame of the SINR medium has to be shorter than 2^32 us");
    const size_t total = c->pending.size();
    bool zero_copy = false;
    int staged = -1;
    RM_TRY(ev_flush_append(c)); // (an append left for a drain reads the records of the tick before out of d_tx)
    if (!sinr) RM_HIP(c->d_tx.ensure(std::max<size_t>(total, 1)));
    if (total) {
        // through pinned staging: the copy is asynchronous, the buffer is reused only after its copy has completed
        const int g = c->h_tx_gen;
        c->h_tx_gen ^= 1;
        if (!c->h_tx_ev[g]) RM_HIP(hipEventCreateWithFlags(&c->h_tx_ev[g], hipEventDisableTiming));
        else RM_HIP(hipEventSynchronize(c->h_tx_ev[g]));
        if (c->h_tx_n[g] < total) {
            if (c->h_tx[g]) RM_HIP(hipHostFree(c->h_tx[g]));
            c->h_tx[g] = nullptr;
            c->h_tx_n[g] = 0;
            const size_t want = std::max<size_t>(total + total / 2, 1024);
            RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_tx[g]), want * sizeof(rm_tx_record), hipHostMallocMapped));
            c->h_tx_n[g] = want;
        }
        std::memcpy(c->h_tx[g], c->pending.data(), total * sizeof(rm_tx_record));
        // A tick of a few thousand frames: the kernels read the records straight from this pinned, host-mapped block (every
        // frame's workgroup fetches its own 64 bytes and leaves them in device memory for whoever comes later) -- a copy
        // engine's ~10 us hand-over per tick is most of what the transfer costs.  Larger ticks take the copy; the SINR
        // medium's records are copied to the tail of its on-air window.
        static const bool no_zero_copy = std::getenv("RM_NO_ZERO_COPY") != nullptr;
        zero_copy = !sinr && !no_zero_copy && total <= 8192;
        if (!zero_copy && !sinr) RM_HIP(hipMemcpyAsync(c->d_tx.p, c->h_tx[g], total * sizeof(rm_tx_record), hipMemcpyHostToDevice, c->stream));
        c->host_src = zero_copy ? c->h_tx[g] : nullptr;
        staged = g;
    }
    int rc;
    if (sinr) {
        int64_t latest_end = INT64_MIN;
        for (const rm_tx_record &r : c->pending) latest_end = std::max(latest_end, r.start_us + r.air_us);
        rc = air_tick_device(c, c->t_begin, nullptr, staged >= 0 ? c->h_tx[staged] : nullptr, int32_t(total), 0, 0, latest_end, true);
    } else {
        rc = run_tick(c, c->d_tx.p, int(total), 0, nullptr, 0, 0, kAirNone, 0);
    }
    c->host_src = nullptr;
    if (staged >= 0) RM_HIP(hipEventRecord(c->h_tx_ev[staged], c->stream)); // (the block is rewritten only after whatever read it)
    c->tick_frac_records = false;
    if (rc != RM_OK) {
        c->air.valid = false;
        return rc;
    }
    c->pending.clear();
    return RM_OK;
}

int result_device(rm_context *c, TickSlot &ts, rm_device_result *out)
{
    if (!ts.have_result) return fail(RM_ERR_STATE, "no evaluated tick");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(materialize(c, ts));
    out->count = ts.last.out_count;
    out->pkt_offset = ts.d_slot_off.p + ts.last.shift;
    out->pkt = ts.d_out_pkt.p;
    out->dst = ts.d_out_dst.p;
    out->verdict = ts.d_out_verdict.p;
    out->rssi = ts.d_out_rssi.p;
    out->sinr = ts.last.out_sinr; // NULL without the SINR extension
    out->capacity = c->cap;
    return RM_OK;
}

int result_count(rm_context *c, TickSlot &ts, uint32_t *count, uint32_t *dropped)
{
    if (!ts.have_result) return fail(RM_ERR_STATE, "no evaluated tick");
    RM_HIP(hipSetDevice(c->device));
    if (!ts.dense_pending) RM_TRY(materialize(c, ts)); // (a dense tick's totals do not wait for its records,
    else RM_TRY(dense_layout(c, ts));                  //  only for its cells' layout)
    uint32_t oc[5];
    RM_HIP(hipMemcpyAsync(oc, ts.last.out_count, sizeof(oc), hipMemcpyDeviceToHost, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    if (oc[4]) return fail(RM_ERR_STATE, record_flag_message(oc[4]));
    if (count) *count = oc[2];
    if (dropped) *dropped = oc[1];
    return RM_OK;
}

} // namespace rmh

extern "C" {

int rm_tick_begin(rm_context *c, int64_t t_begin_us, int64_t t_end_us)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    c->t_begin = t_begin_us;
    c->t_end = t_end_us;
    c->pending.clear();
    c->tick_frac_records = false;
    if (!is_sinr(c)) c->air.valid = false; // (the SINR medium's frames of earlier ticks stay on the device: air_tick_device)
    c->in_tick = true;
    return RM_OK;
}

int rm_enqueue_tx(rm_context *c, int32_t src, int64_t start_us, int64_t air_us, const double *txpower,
                  const int32_t *channel)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    if (!c->in_tick) return fail(RM_ERR_STATE, "rm_enqueue_tx outside rm_tick_begin / rm_tick_flush");
    if (src < 0 || src >= c->n) return fail(RM_ERR_INVALID, "could not find source node"); // SimulatorJSONHandler.java:75-77
    if (air_us < 0) return fail(RM_ERR_INVALID, "negative air time");
    c->pending.push_back(make_record(c, src, start_us, air_us, txpower, channel));
    return RM_OK;
}

int rm_enqueue_tx_records(rm_context *c, const rm_tx_record *recs, int32_t n)
{
    if (!c || n < 0 || (n > 0 && !recs)) return fail(RM_ERR_INVALID, "bad arguments");
    if (!c->in_tick) return fail(RM_ERR_STATE, "rm_enqueue_tx_records outside a tick");
    for (int i = 0; i < n; ++i) {
        if (recs[i].src >= c->n) return fail(RM_ERR_INVALID, "record source out of range");
        // the Tx draw of UDGMRadioMedium.java:87-92 follows the RECORD's txProbability, whatever the node table says
        if (frac(recs[i].txprob)) c->tick_frac_records = true;
        c->pending.push_back(recs[i]);
    }
    return RM_OK;
}

int rm_tick_run(rm_context *c)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    return tick_run_host(c);
}

int rm_tick_flush_view(rm_context *c, rm_host_result *out)
{
    if (!c || !out) return fail(RM_ERR_INVALID, "NULL argument");
    RM_TRY(tick_run_host(c));
    rm::HostView v{};
    RM_TRY(pack_to_stage(c, *c, &v));
    out->count = v.hdr->stored;
    out->n_packets = v.hdr->n_packets;
    out->pkt_offset = v.pkt_offset;
    out->pkt_interference = v.pkt_interference;
    out->pkt = nullptr; // (ABI version 3: pkt_offset says it all)
    out->dst = v.dst;
    out->verdict = v.verdict;
    out->rssi = v.rssi;                                // NULL for the reference's media: ...
    out->pkt_rssi = v.rssi ? nullptr : v.pkt_rssi;     // ... their links carry the packet's transmit power
    out->sinr = c->last.out_sinr ? v.sinr : nullptr; // written by the SINR extension only
    return stage_status(c, v);
}

int rm_tick_flush(rm_context *c, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr,
                  uint32_t cap, uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    RM_TRY(tick_run_host(c));
    rm::HostView v{};
    RM_TRY(pack_to_stage(c, *c, &v));
    if (v.hdr->span_flag) return stage_status(c, v);
    if (count) *count = v.hdr->total;
    const uint32_t k = std::min(v.hdr->stored, cap);
    if (k) {
        if (pkt) { // the packet column is not in the block any more: written here from the offsets
            const uint32_t np_ = v.hdr->n_packets;
            for (uint32_t q = 0; q < np_; ++q)
                for (uint32_t i = v.pkt_offset[q], e = std::min(v.pkt_offset[q + 1], k); i < e; ++i) pkt[i] = int32_t(q);
        }
        if (dst) std::memcpy(dst, v.dst, k * sizeof(int32_t));
        if (verdict) std::memcpy(verdict, v.verdict, k);
        if (rssi && v.rssi) std::memcpy(rssi, v.rssi, k * sizeof(double));
        else if (rssi) // one rssi per packet crossed the link: the caller's array is filled from the offsets
            for (uint32_t q = 0; q < v.hdr->n_packets; ++q)
                for (uint32_t i = v.pkt_offset[q], e = std::min(v.pkt_offset[q + 1], k); i < e; ++i) rssi[i] = v.pkt_rssi[q];
        if (sinr && c->last.out_sinr) std::memcpy(sinr, v.sinr, k * sizeof(double));
        else if (sinr) std::memset(sinr, 0, k * sizeof(double));
    }
    const uint32_t np = v.hdr->n_packets;
    if (pkt_interference && np) std::memcpy(pkt_interference, v.pkt_interference, np);
    if (pkt_offset) std::memcpy(pkt_offset, v.pkt_offset, (size_t(np) + 1) * sizeof(uint32_t));
    RM_TRY(stage_status(c, v));
    if (v.hdr->total > cap) return fail(RM_ERR_CAPACITY, "caller buffers too small for the heard links");
    return RM_OK;
}

int rm_transmit(rm_context *c, int32_t src, int64_t start_us, int64_t hex_length, const double *txpower,
                const int32_t *channel, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr, uint32_t cap,
                uint32_t *count, uint8_t *interference)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    if (src < 0 || src >= c->n) return fail(RM_ERR_INVALID, "could not find source node");
    if (hex_length < 0) return fail(RM_ERR_INVALID, "negative packet length");
    const bool draws_need_exchange = maybe_draws(c) && part_count(c) != c->n;
    if (is_sinr(c) || draws_need_exchange) {
        // the on-air list of earlier calls / the per-rank draw exchange: the general tick path
        RM_TRY(rm_tick_begin(c, start_us, start_us));
        RM_TRY(rm_enqueue_tx(c, src, start_us, rm_air_time_us(hex_length), txpower, channel));
        return rm_tick_flush(c, nullptr, dst, verdict, rssi, sinr, cap, count, interference, nullptr);
    }
    // One packet, no state besides the generator: record in through the kernel arguments, links out
    // through one host-mapped block, one synchronisation.
    RM_HIP(hipSetDevice(c->device));
    c->t_begin = c->t_end = start_us;
    c->in_tick = false;
    c->pending.clear();
    c->air.valid = false;
    if (!c->h_transmit) {
        RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_transmit), sizeof(rm::TransmitResult), hipHostMallocMapped));
        std::memset(c->h_transmit, 0, sizeof(rm::TransmitResult));
    }
    const rm_tx_record rec = make_record(c, src, start_us, rm_air_time_us(hex_length), txpower, channel);
    // Geometric media on a sorted table: the whole packet in one launch of one workgroup.
    RM_TRY(prepare_nodes(c));
    {
        const rm::ModelDev m = model_dev(c);
        const bool f64_filter = c->f32_slack > 0.05 || (m.geo_cut > 0 && c->f32_slack > 0.05 * m.geo_cut);
        // (with the reception stage on, the packet's links have to stay on the device: the tick path)
        if (is_geometric(c) && c->rx_sorted && !f64_filter && c->n_rx > 0 && !c->ev.on && std::getenv("RM_NO_ONE_LAUNCH") == nullptr) {
            if (!c->d_rng.p) RM_TRY(rm_seed(c, 0));
            c->have_result = false; // the links go to the caller only
            const uint32_t seq = ++c->transmit_seq;
            RM_HIP(rm::launch_transmit_one(c->stream, nodes_dev(c), m, rec, c->d_rng.p, c->h_transmit, seq));
            // The kernel publishes `seq` in the host-mapped block after everything else: polling it for
            // the ~10 us the kernel takes beats the wake-up latency of a stream synchronisation; a stream
            // synchronisation still bounds the wait.
            {
                volatile const uint32_t *flag = &c->h_transmit->seq;
                bool seen = false;
                for (int spin = 0; spin < 200000 && !seen; ++spin) seen = (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq);
                if (!seen) RM_HIP(hipStreamSynchronize(c->stream));
            }
            const rm::TransmitResult &r1 = *c->h_transmit;
            if (r1.total != rm::kTransmitFallback) {
                if (count) *count = r1.total;
                if (interference) *interference = uint8_t(r1.interference);
                const uint32_t k1 = std::min(r1.stored, cap);
                if (k1) {
                    if (dst) std::memcpy(dst, r1.dst, k1 * sizeof(int32_t));
                    if (verdict) std::memcpy(verdict, r1.verdict, k1);
                    if (rssi) std::memcpy(rssi, r1.rssi, k1 * sizeof(double));
                    if (sinr) std::memcpy(sinr, r1.sinr, k1 * sizeof(double));
                }
                if (r1.total > cap) return fail(RM_ERR_CAPACITY, "caller buffers too small for the heard links");
                return RM_OK;
            }
            // the kernel declined (unbounded range or more links than its LDS lists hold) and changed nothing
        }
    }
    RM_TRY(ev_flush_append(c));
    RM_HIP(c->d_tx.ensure(1));
    RM_HIP(rm::launch_store_record(c->stream, rec, c->d_tx.p));
    RM_TRY(run_tick(c, c->d_tx.p, 1, 0));
    if (part_count(c) <= 0) { // no receivers in this partition
        if (count) *count = 0;
        if (interference) *interference = 0;
        return RM_OK;
    }
    RM_TRY(materialize(c, *c));
    RM_HIP(rm::launch_pack_result(c->stream, c->last, c->h_transmit));
    RM_HIP(hipStreamSynchronize(c->stream));
    const rm::TransmitResult &r = *c->h_transmit;
    if (r.total > uint32_t(rm::kTransmitMax)) // more links than the block holds: the general copy-out
        return copy_out(c, *c, nullptr, dst, verdict, rssi, sinr, cap, count, interference, nullptr);
    if (count) *count = r.total;
    if (interference) *interference = uint8_t(r.interference);
    const uint32_t k = std::min(r.stored, cap);
    if (k) {
        if (dst) std::memcpy(dst, r.dst, k * sizeof(int32_t));
        if (verdict) std::memcpy(verdict, r.verdict, k);
        if (rssi) std::memcpy(rssi, r.rssi, k * sizeof(double));
        if (sinr) std::memcpy(sinr, r.sinr, k * sizeof(double));
    }
    if (r.dropped) {
        c->air.valid = false;
        return fail(RM_ERR_CAPACITY, "heard links exceed the context's link capacity (rm_set_link_capacity)");
    }
    if (r.total > cap) return fail(RM_ERR_CAPACITY, "caller buffers too small for the heard links");
    return RM_OK;
}

int rm_pack_tx_device_on(rm_context *c, void *hip_stream, const int32_t *dev_src, int32_t n, int64_t start_us,
                         int64_t air_us, rm_tx_record *dev_out)
{
    if (!c || n < 0 || (n > 0 && (!dev_src || !dev_out))) return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    RM_HIP(rm::launch_pack_tx(static_cast<hipStream_t>(hip_stream), nodes_dev(c), dev_src, n, start_us, air_us, dev_out));
    return RM_OK;
}

int rm_pack_tx_batch_device_on(rm_context *c, void *hip_stream, const int32_t *dev_src, int32_t n_ticks, int32_t n,
                               const int64_t *start_us, int64_t air_us, rm_tx_record *dev_out)
{
    if (!c || n < 0 || n_ticks < 1 || n_ticks > RM_MAX_BATCH || !start_us || (n > 0 && (!dev_src || !dev_out)))
        return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    RM_HIP(rm::launch_pack_tx_batch(static_cast<hipStream_t>(hip_stream), nodes_dev(c), dev_src, n_ticks, n, start_us, air_us,
                                    dev_out));
    return RM_OK;
}

int rm_pack_tx_device(rm_context *c, const int32_t *dev_src, int32_t n, int64_t start_us, int64_t air_us,
                      rm_tx_record *dev_out)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    return rm_pack_tx_device_on(c, c->stream, dev_src, n, start_us, air_us, dev_out);
}

int rm_tick_run_device(rm_context *c, int64_t t_begin_us, int64_t t_end_us, const rm_tx_record *dev_new, int32_t n_new)
{
    if (!c || n_new < 0 || (n_new > 0 && !dev_new)) return fail(RM_ERR_INVALID, "bad arguments");
    if (is_sinr(c))
        return fail(RM_ERR_STATE, "the SINR medium keeps frames on the air: records in device memory go through "
                                  "rm_tick_run_records_device, which is told how long they stay (latest_end_us)");
    RM_HIP(hipSetDevice(c->device));
    c->t_begin = t_begin_us;
    c->t_end = t_end_us;
    c->dev_records_from_caller = true;
    const int rc = run_tick(c, dev_new, n_new, 0);
    c->dev_records_from_caller = false;
    return rc;
}

int rm_tick_run_sources_device(rm_context *c, int64_t t_begin_us, int64_t t_end_us, const int32_t *dev_src, int32_t n,
                               int64_t start_us, int64_t air_us)
{
    if (!c || n < 0 || (n > 0 && !dev_src) || air_us < 0) return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    c->t_begin = t_begin_us;
    c->t_end = t_end_us;
    if (!is_sinr(c)) {
        RM_TRY(ev_flush_append(c));
        RM_HIP(c->d_tx.ensure(std::max(n, 1)));
        // (the closed loop's call: rm_events_process follows at once and takes this tick's append into its first launch)
        return run_tick(c, c->d_tx.p, n, 0, dev_src, start_us, air_us, kAirNone, 0, true);
    }
    return air_tick_device(c, t_begin_us, dev_src, nullptr, n, start_us, air_us, start_us + air_us, false);
}

} // extern "C"

namespace rmh {

// The on-air window [air_head, air_tail) of d_air.  Whole batches leave it when their last frame has left the air
// (rm_tick_begin's rule: start + air > t_begin stays).
int air_window_expire(rm_context *c, int64_t t_begin_us)
{
    if (c->air_batches.empty()) c->air_max_t_begin = INT64_MIN;
    if (c->air_batches.empty()) c->air_culled = false; // (nothing on the air that was selected for a region)
    {
        bool fifo = true; // live batches form a suffix of the window?
        size_t first_live = c->air_batches.size();
        for (size_t i = 0; i < c->air_batches.size(); ++i) {
            const bool live = c->air_batches[i].end_us > t_begin_us;
            if (live && first_live == c->air_batches.size()) first_live = i;
            if (!live && first_live != c->air_batches.size()) fifo = false;
        }
        if (fifo) {
            for (size_t i = 0; i < first_live; ++i) c->air_head += size_t(c->air_batches[i].count);
            c->air_batches.erase(c->air_batches.begin(), c->air_batches.begin() + first_live);
        } else { // an earlier batch outlives a later one: compact the live batches to the front
            DevBuf<rm_tx_record> fresh;
            RM_HIP(fresh.ensure(std::max<size_t>(c->d_air.n, 1)));
            size_t src = c->air_head, dst = 0;
            std::vector<rm_context::AirBatch> keep;
            for (const auto &bt : c->air_batches) {
                if (bt.end_us > t_begin_us) {
                    RM_HIP(hipMemcpyAsync(fresh.p + dst, c->d_air.p + src, size_t(bt.count) * sizeof(rm_tx_record),
                                          hipMemcpyDeviceToDevice, c->stream));
                    dst += size_t(bt.count);
                    keep.push_back(bt);
                }
                src += size_t(bt.count);
            }
            RM_HIP(hipStreamSynchronize(c->stream));
            c->d_air.release();
            c->d_air = fresh;
            c->air_batches.swap(keep);
            c->air_head = 0;
            c->air_tail = dst;
        }
    }
    return RM_OK;
}

// room for n more frames at the window's tail; the window slides to the front when the buffer is used up
int air_window_reserve(rm_context *c, size_t n_more)
{
    const size_t n = n_more;
    const size_t live = c->air_tail - c->air_head;
    if (c->air_tail + size_t(n) > c->d_air.n) {
        // The window slides to the front of the OTHER of two buffers: one copy on the context's stream -- ordered behind every
        // kernel that still reads either buffer -- and no allocation, no synchronisation in the steady state (a steady stream of
        // batches used to pay a hipMalloc, a stream synchronisation and a hipFree every third batch: the device ran dry each time).
        const size_t want = std::max<size_t>(4 * (live + size_t(n)), 1 << 16);
        if (c->d_air_alt.n < want) {
            RM_HIP(hipStreamSynchronize(c->stream)); // (the buffer about to be replaced may still be read by kernels in flight)
            RM_HIP(c->d_air_alt.ensure(std::max(want, c->d_air.n)));
        }
        if (live)
            RM_HIP(hipMemcpyAsync(c->d_air_alt.p, c->d_air.p + c->air_head, live * sizeof(rm_tx_record), hipMemcpyDeviceToDevice, c->stream));
        std::swap(c->d_air, c->d_air_alt);
        c->air_head = 0;
        c->air_tail = live;
    }
    return RM_OK;
}

// The SINR medium's tick (rm_host.hpp) -- its new frames built from source indices (dev_src: all with the same start and air
// time) or given as records (dev_new, in device memory or in the host's pinned staging block; `latest_end_us` bounds their
// start + air: records in device memory are never read by the host).
// The frames of earlier calls that are still on the air stay resident on the device (the window [air_head, air_tail) of
// d_air): a tick that only adds frames sweeps the new ones, a rebuild of the on-air lists sweeps the whole window.
int air_tick_device(rm_context *c, int64_t t_begin_us, const int32_t *dev_src, const rm_tx_record *dev_new, int32_t n, int64_t start_us,
                    int64_t air_us, int64_t latest_end_us, bool new_on_host)
{
    RM_TRY(air_window_expire(c, t_begin_us));
    RM_TRY(air_window_reserve(c, size_t(n)));
    const size_t live = c->air_tail - c->air_head;
    if (dev_src && air_us > int64_t(UINT32_MAX)) return fail(RM_ERR_INVALID, "a frame of the SINR medium has to be shorter than 2^32 us");
    if (dev_new && n > 0) // the caller's records join the window (every later tick looks at them while they are on the air)
        RM_HIP(hipMemcpyAsync(c->d_air.p + c->air_tail, dev_new, size_t(n) * sizeof(rm_tx_record),
                              new_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
    const bool clock_back = t_begin_us < c->air_max_t_begin;
    c->air_max_t_begin = std::max(c->air_max_t_begin, t_begin_us);
    uint32_t oldest = 0;
    bool unknown = false;
    for (const auto &bt : c->air_batches) {
        unknown = unknown || bt.tick == 0;
        oldest = (oldest == 0 || bt.tick < oldest) ? bt.tick : oldest;
    }
    RM_TRY(prepare_nodes(c)); // (a changed table makes the lists stale, and decides whether the tick can go by scan)
    const int air_mode = air_scan_applies(c, n) ? kAirScan
                                                : ((!unknown && air_lists_current(c, t_begin_us, oldest)) ? kAirIncremental : kAirRebuild);
    // rm_tick_begin's rule is applied tick by tick: a frame that had left the air when a tick began does not come back when
    // the clock does.  The kernels compare with THIS tick's t_begin, so when the clock goes back such frames are first
    // retired in the window for good; a rebuild of the lists retires them too, so that it does not sweep frames (inside
    // batches that are still alive) that have nothing to say any more.
    if (live > 0 && (clock_back || air_mode == kAirRebuild))
        RM_HIP(rm::launch_air_expire(c->stream, c->d_air.p + c->air_head, int(live), c->air_max_t_begin));
    // the records of the new frames are built at the window's tail either way; an incremental tick sweeps only those
    const int first_new = (air_mode == kAirIncremental) ? 0 : int(live);
    const rm_tx_record *base = c->d_air.p + c->air_head + (air_mode == kAirIncremental ? live : 0);
    const int rc = run_tick(c, base, first_new + n, first_new, dev_src, start_us, air_us, air_mode, oldest);
    if (rc != RM_OK) {
        c->air.valid = false;
        return rc;
    }
    // the frames on the air were selected for this partition's region: are its receivers still inside what they were selected for?
    if (c->air_culled && live > 0 && c->d_cull_ring.p && c->last.stage_count)
        RM_HIP(rm::launch_cull_check(c->stream, nodes_dev(c), c->d_cull_ring.p, t_begin_us, c->last.stage_count + 6));
    if (air_mode == kAirRebuild)
        for (auto &bt : c->air_batches) bt.tick = c->air.tick;
    c->air_tail += size_t(n);
    if (n > 0) c->air_batches.push_back({n, latest_end_us, air_mode == kAirScan ? 0u : c->air.tick}); // (0: not in the lists)
    return RM_OK;
}

} // namespace rmh

extern "C" {

int rm_tick_run_records_device(rm_context *c, int64_t t_begin_us, int64_t t_end_us, const rm_tx_record *dev_new, int32_t n_new,
                               int64_t latest_end_us)
{
    if (!c || n_new < 0 || (n_new > 0 && !dev_new)) return fail(RM_ERR_INVALID, "bad arguments");
    if (!is_sinr(c)) return rm_tick_run_device(c, t_begin_us, t_end_us, dev_new, n_new);
    RM_HIP(hipSetDevice(c->device));
    c->t_begin = t_begin_us;
    c->t_end = t_end_us;
    return air_tick_device(c, t_begin_us, nullptr, dev_new, n_new, 0, 0, latest_end_us, false);
}

// The last tick's heard links as the dense tick leaves them (rm_dense.hip): per (packet, chunk of 1024 consecutive nodes) cell
// sixteen lane masks -- bit l of mask k: node rx_first + 1024 * chunk + 64 * k + l heard the packet -- and the cell's count.
int rm_result_dense(rm_context *c, rm_dense_result *out)
{
    if (!c || !out) return fail(RM_ERR_INVALID, "NULL argument");
    if (!c->have_result || !c->dense_result) return fail(RM_ERR_STATE, "the last tick did not take the dense form (rm_result_device has its records)");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(dense_layout(c, *c)); // (pkt_offset and count: laid out when the first reader asks)
    out->cell_mask = c->d_dense_mask.p;
    out->cell_count = c->d_cnt.p;
    out->count = c->last.out_count + 2;
    out->pkt_offset = c->d_slot_off.p + c->last.shift;
    out->pkt_interference = c->d_pkt_interf.p;
    out->n_packets = c->last_n_new;
    out->chunks = c->dense_chunks;
    out->rx_first = c->dense_rx_first;
    return RM_OK;
}

int rm_result_device(rm_context *c, rm_device_result *out)
{
    if (!c || !out) return fail(RM_ERR_INVALID, "NULL argument");
    return result_device(c, *c, out);
}

int rm_draws_pending(const rm_context *c) { return (c && c->draws_pending) ? 1 : 0; }

int rm_draw_counts_device(rm_context *c, const uint32_t **dev_counts, int32_t *n_new)
{
    if (!c || !dev_counts) return fail(RM_ERR_INVALID, "NULL argument");
    if (!c->have_result) return fail(RM_ERR_STATE, "no evaluated tick");
    *dev_counts = c->d_pkt_draw_cnt.p;
    if (n_new) *n_new = c->last_n_new;
    return RM_OK;
}

int rm_draw_counts_to(rm_context *c, uint32_t *dev_out)
{
    if (!c || !dev_out) return fail(RM_ERR_INVALID, "NULL argument");
    if (!c->have_result) return fail(RM_ERR_STATE, "no evaluated tick");
    RM_HIP(hipSetDevice(c->device));
    if (c->last_n_new > 0)
        RM_HIP(hipMemcpyAsync(dev_out, c->d_pkt_draw_cnt.p, size_t(c->last_n_new) * 4, hipMemcpyDeviceToDevice, c->stream));
    return RM_OK;
}

int rm_draw_nodes_device(rm_context *c, const int32_t **dev_nodes)
{
    if (!c || !dev_nodes) return fail(RM_ERR_INVALID, "NULL argument");
    if (!c->have_result || !c->draws_pending || !part_spatial(c))
        return fail(RM_ERR_STATE, "no tick of a spatial partition is waiting for the draw exchange");
    *dev_nodes = c->d_draw_nodes.p;
    return RM_OK;
}

int rm_tick_finish_draws_nodes(rm_context *c, const uint32_t *all_counts, const int32_t *all_nodes, uint32_t stride, int32_t world,
                               int on_device)
{
    if (!c || !all_counts || world < 1 || (stride > 0 && !all_nodes)) return fail(RM_ERR_INVALID, "bad arguments");
    if (!c->draws_pending) return fail(RM_ERR_STATE, "no tick is waiting for draw counts");
    RM_HIP(hipSetDevice(c->device));
    const int n_new = c->last_n_new;
    const uint32_t *dev_cnt = all_counts;
    const int32_t *dev_nodes = all_nodes;
    if (!on_device) {
        RM_HIP(c->d_all_cnt.ensure(size_t(world) * std::max(n_new, 1)));
        RM_HIP(c->d_all_nodes.ensure(std::max<size_t>(size_t(world) * stride, 1)));
        RM_HIP(hipMemcpyAsync(c->d_all_cnt.p, all_counts, size_t(world) * n_new * 4, hipMemcpyHostToDevice, c->stream));
        if (stride) RM_HIP(hipMemcpyAsync(c->d_all_nodes.p, all_nodes, size_t(world) * stride * 4, hipMemcpyHostToDevice, c->stream));
        RM_HIP(hipStreamSynchronize(c->stream)); // the caller's buffers may go away
        dev_cnt = c->d_all_cnt.p;
        dev_nodes = c->d_all_nodes.p;
    }
    RM_HIP(c->d_all_off.ensure(size_t(world) * std::max(n_new, 1)));
    RM_HIP(rm::launch_draws_apply_nodes(c->stream, c->pending_model, c->last, dev_cnt, c->d_all_off.p, dev_nodes, stride, world));
    c->draws_pending = false;
    if (c->ev.on) RM_TRY(ev_append(c, *c)); // the verdicts are final now
    return RM_OK;
}

int rm_tick_finish_draws(rm_context *c, const uint32_t *all_counts, int32_t world, int32_t rank, int on_device)
{
    if (!c || !all_counts || world < 1 || rank < 0 || rank >= world) return fail(RM_ERR_INVALID, "bad arguments");
    if (!c->draws_pending) return fail(RM_ERR_STATE, "no tick is waiting for draw counts");
    if (part_spatial(c))
        return fail(RM_ERR_STATE, "a spatial partition's draws interleave with the other ranks': exchange rm_draw_nodes_device as "
                                  "well and call rm_tick_finish_draws_nodes");
    RM_HIP(hipSetDevice(c->device));
    const uint32_t *dev = all_counts;
    if (!on_device) {
        const size_t n = size_t(world) * std::max(c->last_n_new, 1);
        RM_HIP(c->d_all_cnt.ensure(n));
        RM_HIP(hipMemcpyAsync(c->d_all_cnt.p, all_counts, size_t(world) * c->last_n_new * 4, hipMemcpyHostToDevice, c->stream));
        RM_HIP(hipStreamSynchronize(c->stream)); // the caller's buffer may go away
        dev = c->d_all_cnt.p;
    }
    RM_HIP(rm::launch_draws_apply(c->stream, c->pending_model, c->last, dev, world, rank));
    c->draws_pending = false;
    if (c->ev.on) RM_TRY(ev_append(c, *c)); // the verdicts are final now
    return RM_OK;
}

int rm_result_copy(rm_context *c, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr, uint32_t cap,
                   uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    if (!c->have_result) return fail(RM_ERR_STATE, "no evaluated tick");
    RM_HIP(hipSetDevice(c->device));
    return copy_out(c, *c, pkt, dst, verdict, rssi, sinr, cap, count, pkt_interference, pkt_offset);
}

int rm_result_count(rm_context *c, uint32_t *count, uint32_t *dropped)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    return result_count(c, *c, count, dropped);
}

} // extern "C"
