// rm_api_group.cpp -- C ABI: several devices (or partitions of one) behind one host thread (rm_group_*).
#include "rm_host.hpp"

using namespace rmh;

struct rm_group {
    std::vector<rm_context *> m;
    int n_nodes = 0;
    int n_new = 0; // frames of the running tick
    bool in_tick = false;
    bool spatial = true;          // members own regions of the plane (rm_set_partition_spatial), not ranges of node indices
    std::vector<uint32_t> counts; // [world][n_new] per-packet draw counts
    std::vector<int32_t> draw_nodes; // [world][stride] spatial partitions: the node of every drawing link, packet-major
    // the device-resident tick (rm_group_tick_run_sources_device): one RCCL communicator over the members' devices
    // (ncclCommInitAll), or -- several members on ONE device, where RCCL has no second rank to offer -- copies on that device
    int comm_state = 0;           // 0 not tried, 1 RCCL, 2 device copies
    std::vector<hipEvent_t> packed; // per member: its frames are packed (device-copy mode)
};

extern "C" {

int rm_group_create(int32_t n_members, const int32_t *device_ordinals, rm_group **out)
{
    if (!out || n_members < 1 || !device_ordinals) return fail(RM_ERR_INVALID, "bad arguments");
    *out = nullptr;
    std::unique_ptr<rm_group> g(new rm_group());
    for (int i = 0; i < n_members; ++i) {
        rm_context *c = nullptr;
        const int rc = rm_create(device_ordinals[i], &c);
        if (rc != RM_OK) {
            for (rm_context *k : g->m) rm_destroy(k);
            return rc;
        }
        g->m.push_back(c);
    }
    *out = g.release();
    return RM_OK;
}

void rm_group_destroy(rm_group *g)
{
    if (!g) return;
    for (hipEvent_t e : g->packed) (void)hipEventDestroy(e);
    for (rm_context *c : g->m) rm_destroy(c);
    delete g;
}

int rm_group_set_partitioning(rm_group *g, int32_t spatial)
{
    if (!g) return fail(RM_ERR_INVALID, "group is NULL");
    if (g->n_nodes > 0 && (spatial != 0) != g->spatial) return fail(RM_ERR_STATE, "choose the partitioning before rm_group_nodes_upload");
    g->spatial = spatial != 0;
    return RM_OK;
}

int rm_group_size(const rm_group *g) { return g ? int(g->m.size()) : fail(RM_ERR_INVALID, "group is NULL"); }

rm_context *rm_group_context(rm_group *g, int32_t member)
{
    if (!g || member < 0 || size_t(member) >= g->m.size()) return nullptr;
    return g->m[size_t(member)];
}

#define RM_GROUP_ALL(call)                                                                             \
    do {                                                                                               \
        if (!g) return fail(RM_ERR_INVALID, "group is NULL");                                          \
        for (rm_context *c : g->m) RM_TRY(call);                                                       \
        return RM_OK;                                                                                  \
    } while (0)

int rm_group_set_model(rm_group *g, const rm_model_params *p) { RM_GROUP_ALL(rm_set_model(c, p)); }
int rm_group_set_n2n_matrix(rm_group *g, int32_t m, const double *row_major) { RM_GROUP_ALL(rm_set_n2n_matrix(c, m, row_major)); }
int rm_group_seed(rm_group *g, int64_t seed) { RM_GROUP_ALL(rm_seed(c, seed)); }
int rm_group_set_link_capacity(rm_group *g, uint32_t max_links) { RM_GROUP_ALL(rm_set_link_capacity(c, max_links)); }
int rm_group_set_time(rm_group *g, int64_t t) { RM_GROUP_ALL(rm_set_time(c, t)); }
int rm_group_node_update(rm_group *g, int32_t node, double x, double y, double z, double txpower, int32_t channel,
                         uint8_t enabled, double rxprob, double txprob)
{
    RM_GROUP_ALL(rm_node_update(c, node, x, y, z, txpower, channel, enabled, rxprob, txprob));
}

int rm_group_get_rng_state(rm_group *g, uint64_t *state48)
{
    if (!g || g->m.empty()) return fail(RM_ERR_INVALID, "group is NULL");
    return rm_get_rng_state(g->m[0], state48); // every member walks the same generator
}

int rm_group_nodes_upload(rm_group *g, int32_t n, const double *x, const double *y, const double *z, const double *txpower,
                          const int32_t *channel, const uint8_t *enabled, const double *rxprob, const double *txprob,
                          const int32_t *int_id)
{
    if (!g) return fail(RM_ERR_INVALID, "group is NULL");
    const int64_t world = int64_t(g->m.size());
    for (int64_t r = 0; r < world; ++r) {
        rm_context *c = g->m[size_t(r)];
        RM_TRY(rm_nodes_upload(c, n, x, y, z, txpower, channel, enabled, rxprob, txprob, int_id));
        const int32_t lo = int32_t(int64_t(n) * r / world), hi = int32_t(int64_t(n) * (r + 1) / world);
        if (g->spatial) RM_TRY(rm_set_partition_spatial(c, int32_t(r), int32_t(world))); // receivers partitioned by region
        else RM_TRY(rm_set_partition(c, lo, hi - lo));                                   // ... or by node index range
    }
    g->n_nodes = n;
    return RM_OK;
}

int rm_group_tick_begin(rm_group *g, int64_t t_begin_us, int64_t t_end_us)
{
    if (!g) return fail(RM_ERR_INVALID, "group is NULL");
    for (rm_context *c : g->m) RM_TRY(rm_tick_begin(c, t_begin_us, t_end_us));
    g->n_new = 0;
    g->in_tick = true;
    return RM_OK;
}

int rm_group_enqueue_tx(rm_group *g, int32_t src, int64_t start_us, int64_t air_us, const double *txpower, const int32_t *channel)
{
    if (!g || !g->in_tick) return fail(RM_ERR_STATE, "rm_group_enqueue_tx outside a tick");
    for (rm_context *c : g->m) RM_TRY(rm_enqueue_tx(c, src, start_us, air_us, txpower, channel)); // a "broadcast": the host has the record
    g->n_new += 1;
    return RM_OK;
}

int rm_group_enqueue_tx_records(rm_group *g, const rm_tx_record *recs, int32_t n)
{
    if (!g || !g->in_tick) return fail(RM_ERR_STATE, "rm_group_enqueue_tx_records outside a tick");
    for (rm_context *c : g->m) RM_TRY(rm_enqueue_tx_records(c, recs, n));
    g->n_new += n;
    return RM_OK;
}

// probabilistic links: the members' per-packet draw counts (regions: and the drawing links' nodes), through the host
static int group_finish_draws(rm_group *g, int n_new)
{
    const int world = int(g->m.size());
    bool pending = false;
    for (rm_context *c : g->m) pending = pending || c->draws_pending;
    if (pending) {
        g->counts.assign(size_t(world) * size_t(std::max(n_new, 1)), 0u);
        for (int r = 0; r < world; ++r) {
            rm_context *c = g->m[size_t(r)];
            if (!c->draws_pending) return fail(RM_ERR_STATE, "the members disagree on whether their links draw");
            RM_HIP(hipSetDevice(c->device));
            if (n_new > 0)
                RM_HIP(hipMemcpyAsync(g->counts.data() + size_t(r) * n_new, c->d_pkt_draw_cnt.p, size_t(n_new) * 4,
                                      hipMemcpyDeviceToHost, c->stream));
        }
        for (rm_context *c : g->m) {
            RM_HIP(hipSetDevice(c->device));
            RM_HIP(hipStreamSynchronize(c->stream));
        }
        if (!g->spatial || world == 1) {
            for (int r = 0; r < world; ++r) RM_TRY(rm_tick_finish_draws(g->m[size_t(r)], g->counts.data(), world, r, 0));
        } else {
            // regions interleave in node order: the members' lists of drawing nodes go round as well
            std::vector<uint64_t> tot(static_cast<size_t>(world), 0);
            uint64_t stride = 0;
            for (int r = 0; r < world; ++r) {
                for (int q = 0; q < n_new; ++q) tot[size_t(r)] += g->counts[size_t(r) * n_new + q];
                stride = std::max(stride, tot[size_t(r)]);
            }
            g->draw_nodes.assign(size_t(world) * size_t(std::max<uint64_t>(stride, 1)), 0);
            for (int r = 0; r < world; ++r) {
                rm_context *c = g->m[size_t(r)];
                RM_HIP(hipSetDevice(c->device));
                if (tot[size_t(r)])
                    RM_HIP(hipMemcpyAsync(g->draw_nodes.data() + size_t(r) * stride, c->d_draw_nodes.p, size_t(tot[size_t(r)]) * 4,
                                          hipMemcpyDeviceToHost, c->stream));
            }
            for (rm_context *c : g->m) {
                RM_HIP(hipSetDevice(c->device));
                RM_HIP(hipStreamSynchronize(c->stream));
            }
            for (int r = 0; r < world; ++r)
                RM_TRY(rm_tick_finish_draws_nodes(g->m[size_t(r)], g->counts.data(), g->draw_nodes.data(), uint32_t(stride), world, 0));
        }
    }
    return RM_OK;
}

// the members' results in their pinned blocks, merged packet by packet by node index
static int group_merge(rm_group *g, int n_new, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr, uint32_t cap,
                       uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset)
{
    const int world = int(g->m.size());
    std::vector<rm_host_result> res(static_cast<size_t>(world));
    int first_error = RM_OK;
    std::string first_msg;
    for (int r = 0; r < world; ++r) {
        rm_context *c = g->m[size_t(r)];
        RM_HIP(hipSetDevice(c->device));
        rm::HostView v{};
        RM_TRY(pack_to_stage(c, *c, &v));
        rm_host_result &o = res[size_t(r)];
        o.count = v.hdr->stored;
        o.n_packets = v.hdr->n_packets;
        o.pkt_offset = v.pkt_offset;
        o.pkt_interference = v.pkt_interference;
        o.pkt = nullptr;
        o.dst = v.dst;
        o.verdict = v.verdict;
        o.rssi = v.rssi;
        o.pkt_rssi = v.rssi ? nullptr : v.pkt_rssi;
        o.sinr = c->last.out_sinr ? v.sinr : nullptr;
        const int st = stage_status(c, v);
        if (st != RM_OK && first_error == RM_OK) {
            first_error = st;
            first_msg = g_err;
        }
    }
    uint64_t total = 0;
    for (const auto &o : res) total += o.count;
    if (count) *count = uint32_t(std::min<uint64_t>(total, 0xFFFFFFFFu));
    uint32_t w = 0;
    std::vector<uint32_t> cur(static_cast<size_t>(world)), end(static_cast<size_t>(world));
    for (int q = 0; q < n_new; ++q) {
        if (pkt_offset) pkt_offset[q] = w;
        // every member's links of the packet are in node order: a k-way merge by node index restores the reference's
        // visiting order (with index ranges the members' runs simply follow each other)
        for (int r = 0; r < world; ++r) {
            const rm_host_result &o = res[size_t(r)];
            const bool have = uint32_t(q) < o.n_packets;
            cur[size_t(r)] = have ? o.pkt_offset[q] : 0u;
            end[size_t(r)] = have ? std::min(o.pkt_offset[q + 1], o.count) : 0u;
        }
        for (;;) {
            int best = -1;
            for (int r = 0; r < world; ++r)
                if (cur[size_t(r)] < end[size_t(r)] &&
                    (best < 0 || res[size_t(r)].dst[cur[size_t(r)]] < res[size_t(best)].dst[cur[size_t(best)]]))
                    best = r;
            if (best < 0) break;
            const rm_host_result &o = res[size_t(best)];
            const uint32_t i = cur[size_t(best)]++;
            if (w < cap) {
                if (pkt) pkt[w] = q;
                if (dst) dst[w] = o.dst[i];
                if (verdict) verdict[w] = o.verdict[i];
                if (rssi) rssi[w] = o.rssi ? o.rssi[i] : o.pkt_rssi[q];
                if (sinr) sinr[w] = o.sinr ? o.sinr[i] : 0.0;
            }
            ++w;
        }
        // the packet-level Tx-failure flag is the same on every member (one generator, one draw)
        if (pkt_interference && world > 0 && uint32_t(q) < res[0].n_packets) pkt_interference[q] = res[0].pkt_interference[q];
    }
    if (pkt_offset) pkt_offset[std::max(n_new, 0)] = w;
    if (first_error != RM_OK) return fail(first_error, first_msg);
    if (total > cap) return fail(RM_ERR_CAPACITY, "caller buffers too small for the heard links");
    return RM_OK;
}

// How the members exchange their packed frames: RCCL when every member has its own device (or there is one member),
// copies on the device when several members share one (a test configuration: RCCL admits one rank per device).
static int group_comm(rm_group *g)
{
    if (g->comm_state) return RM_OK;
    const int world = int(g->m.size());
    bool distinct = true;
    for (int i = 0; i < world; ++i)
        for (int k = i + 1; k < world; ++k) distinct = distinct && g->m[size_t(i)]->device != g->m[size_t(k)]->device;
    static const bool no_rccl = std::getenv("RM_GROUP_NO_RCCL") != nullptr;
    if (distinct && !no_rccl) {
        RM_TRY(group_comm_init(g->m.data(), world, nullptr));
        g->comm_state = 1;
        return RM_OK;
    }
    for (int i = 0; i < world; ++i) {
        hipEvent_t e = nullptr;
        RM_HIP(hipSetDevice(g->m[size_t(i)]->device));
        RM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g->packed.push_back(e);
        g->m[size_t(i)]->comm_world = world; // (no communicator: the group moves the blocks itself)
        g->m[size_t(i)]->comm_rank = i;
    }
    g->comm_state = 2;
    return RM_OK;
}

int rm_group_uses_rccl(rm_group *g)
{
    if (!g) return fail(RM_ERR_INVALID, "group is NULL");
    RM_TRY(group_comm(g));
    return g->comm_state == 1 ? 1 : 0;
}

int rm_group_tick_run_sources_device(rm_group *g, int64_t t_begin_us, int64_t t_end_us, const int32_t *const *dev_src, int32_t slots,
                                     int64_t start_us, int64_t air_us)
{
    if (!g || !dev_src || slots < 1 || air_us < 0) return fail(RM_ERR_INVALID, "bad arguments");
    RM_TRY(group_comm(g));
    const int world = int(g->m.size());
    const size_t block = size_t(slots) * sizeof(rm_tx_record);
    std::vector<const void *> mine(static_cast<size_t>(world));
    std::vector<void *> all(static_cast<size_t>(world));
    // every member packs the frames of ITS transmitters (dev_src[r]: `slots` node indices on member r's device, -1 = padding)
    for (int r = 0; r < world; ++r) {
        rm_context *c = g->m[size_t(r)];
        if (!dev_src[r]) return fail(RM_ERR_INVALID, "bad arguments");
        RM_HIP(hipSetDevice(c->device));
        RM_HIP(c->d_dist_mine.ensure(size_t(slots)));
        RM_HIP(c->d_dist_all.ensure(size_t(slots) * size_t(world)));
        RM_HIP(rm::launch_pack_tx(c->stream, nodes_dev(c), dev_src[r], slots, start_us, air_us, c->d_dist_mine.p));
        mine[size_t(r)] = c->d_dist_mine.p;
        all[size_t(r)] = c->d_dist_all.p;
        if (g->comm_state == 2) RM_HIP(hipEventRecord(g->packed[size_t(r)], c->stream));
    }
    if (g->comm_state == 1) {
        RM_TRY(group_all_gather(g->m.data(), world, mine.data(), all.data(), block)); // RCCL over xGMI, all ranks in one group call
    } else {
        for (int q = 0; q < world; ++q) {
            rm_context *c = g->m[size_t(q)];
            RM_HIP(hipSetDevice(c->device));
            for (int r = 0; r < world; ++r) {
                if (r != q) RM_HIP(hipStreamWaitEvent(c->stream, g->packed[size_t(r)], 0));
                RM_HIP(hipMemcpyAsync(static_cast<char *>(all[size_t(q)]) + size_t(r) * block, mine[size_t(r)], block,
                                      hipMemcpyDeviceToDevice, c->stream));
            }
        }
    }
    // every member sweeps the gathered frames against its receivers; all launches are enqueued before anything is waited for
    for (rm_context *c : g->m)
        RM_TRY(rm_tick_run_records_device(c, t_begin_us, t_end_us, c->d_dist_all.p, slots * world, start_us + air_us));
    g->n_new = slots * world;
    return group_finish_draws(g, g->n_new);
}

int rm_group_result_copy(rm_group *g, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr, uint32_t cap,
                         uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset)
{
    if (!g) return fail(RM_ERR_INVALID, "group is NULL");
    for (rm_context *c : g->m)
        if (!c->have_result) return fail(RM_ERR_STATE, "no evaluated tick");
    return group_merge(g, g->n_new, pkt, dst, verdict, rssi, sinr, cap, count, pkt_interference, pkt_offset);
}

int rm_group_tick_flush(rm_group *g, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr, uint32_t cap,
                        uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset)
{
    if (!g || !g->in_tick) return fail(RM_ERR_STATE, "rm_group_tick_flush without rm_group_tick_begin");
    g->in_tick = false;
    const int n_new = g->n_new;
    // every member's launches are enqueued before anything is waited for
    for (rm_context *c : g->m) RM_TRY(tick_run_host(c));
    RM_TRY(group_finish_draws(g, n_new));
    return group_merge(g, n_new, pkt, dst, verdict, rssi, sinr, cap, count, pkt_interference, pkt_offset);
}

} // extern "C"
