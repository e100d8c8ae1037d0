// rm_api_nodes.cpp -- C ABI: the node table -- Simulator.getNodes() snapshot, receiver table in engine order, changed nodes, partitions.
#include "rm_host.hpp"

using namespace rmh;

namespace rmh {

// k-d split of items[lo, hi) down to groups of 64: the left part always holds a multiple of 64
// receivers, so every group of 64 consecutive engine positions is one leaf (a compact box).
// The items carry their coordinates (no indirection in the comparisons); ties are broken by the node
// index, so the leaves do not depend on how the work is spread over threads: the first levels hand
// their right halves to new threads.
struct KdItem {
    double v[3];
    int32_t idx;
    int32_t pad;
};

void kd_split(KdItem *items, int lo, int hi, int spawn_levels)
{
    const int cnt = hi - lo;
    if (cnt <= rm::kGroup) return;
    double mn[3], mx[3];
    for (int a = 0; a < 3; ++a) mn[a] = mx[a] = items[lo].v[a];
    for (int i = lo + 1; i < hi; ++i)
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], items[i].v[a]);
            mx[a] = std::max(mx[a], items[i].v[a]);
        }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
        if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
    const int groups = (cnt + rm::kGroup - 1) / rm::kGroup;
    const int mid = lo + ((groups + 1) / 2) * rm::kGroup;
    std::nth_element(items + lo, items + mid, items + hi, [axis](const KdItem &a, const KdItem &b) {
        return a.v[axis] < b.v[axis] || (a.v[axis] == b.v[axis] && a.idx < b.idx);
    });
    if (spawn_levels > 0 && cnt > 8192) {
        std::thread right;
        bool spawned = true;
        try {
            right = std::thread(kd_split, items, mid, hi, spawn_levels - 1);
        } catch (...) { // no thread to be had: this one does both halves
            spawned = false;
        }
        kd_split(items, lo, mid, spawn_levels - 1);
        if (spawned) right.join();
        else kd_split(items, mid, hi, 0);
    } else {
        kd_split(items, lo, mid, 0);
        kd_split(items, mid, hi, 0);
    }
}

template <typename T> int upload_gather(DevBuf<T> &d, const std::vector<T> &src, const std::vector<int32_t> &perm,
                                        hipStream_t s, std::vector<T> &tmp)
{
    tmp.resize(perm.size());
    for (size_t i = 0; i < perm.size(); ++i) tmp[i] = src[perm[i]];
    RM_HIP(d.ensure(std::max<size_t>(tmp.size(), 1)));
    if (!tmp.empty()) RM_HIP(hipMemcpyAsync(d.p, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, s));
    RM_HIP(hipStreamSynchronize(s));
    return RM_OK;
}

// Which region of the plane is part `p` of `parts`?  The top levels of the same k-d split: the node set is cut along its
// widest axis into two parts holding groups of 64 in proportion to the parts they will be cut into, until one part is
// left.  Ties go by node index, so every rank -- each computes this for itself from the same table -- gets the same cut.
// `owner`: instead of following one part, label every node with its part (rm_partition_of_nodes).
static void region_split(KdItem *items, int lo, int hi, int part0, int parts, int want, std::vector<int32_t> *members, int32_t *owner)
{
    if (parts <= 1 || hi - lo <= 0) {
        if (owner)
            for (int i = lo; i < hi; ++i) owner[items[i].idx] = part0;
        if (members && part0 == want)
            for (int i = lo; i < hi; ++i) members->push_back(items[i].idx);
        return;
    }
    const int cnt = hi - lo;
    double mn[3], mx[3];
    for (int a = 0; a < 3; ++a) mn[a] = mx[a] = items[lo].v[a];
    for (int i = lo + 1; i < hi; ++i)
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], items[i].v[a]);
            mx[a] = std::max(mx[a], items[i].v[a]);
        }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
        if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
    const int left_parts = (parts + 1) / 2;
    const int64_t groups = (cnt + rm::kGroup - 1) / rm::kGroup;
    const int64_t left_groups = (groups * left_parts + parts - 1) / parts;
    const int mid = int(std::min<int64_t>(hi, lo + left_groups * rm::kGroup));
    if (mid < hi)
        std::nth_element(items + lo, items + mid, items + hi, [axis](const KdItem &a, const KdItem &b) {
            return a.v[axis] < b.v[axis] || (a.v[axis] == b.v[axis] && a.idx < b.idx);
        });
    if (owner || want < part0 + left_parts) region_split(items, lo, mid, part0, left_parts, want, members, owner);
    if (owner || want >= part0 + left_parts) region_split(items, mid, hi, part0 + left_parts, parts - left_parts, want, members, owner);
}

static std::vector<KdItem> all_items(const rm_context *c)
{
    std::vector<KdItem> items(static_cast<size_t>(c->n));
    for (int i = 0; i < c->n; ++i) items[size_t(i)] = KdItem{{c->x[i], c->y[i], c->z[i]}, i, 0};
    return items;
}

// the member nodes of this context's spatial partition, from the node table as it is now
int select_region(rm_context *c)
{
    c->sp_nodes.clear();
    if (!part_spatial(c)) return RM_OK;
    std::vector<KdItem> items = all_items(c);
    region_split(items.data(), 0, c->n, 0, c->sp_parts, c->sp_part, &c->sp_nodes, nullptr);
    std::sort(c->sp_nodes.begin(), c->sp_nodes.end());
    std::vector<uint8_t> member(size_t(std::max(c->n, 1)), 0);
    for (int32_t i : c->sp_nodes) member[size_t(i)] = 1;
    RM_HIP(c->d_member.ensure(member.size()));
    RM_HIP(hipMemcpyAsync(c->d_member.p, member.data(), member.size(), hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    c->rx_dirty = true;
    return RM_OK;
}

// (re)build the receiver table of the partition in engine order
int rebuild_receivers(rm_context *c)
{
    const int first = part_first(c), count = part_count(c), span = pos_span(c);
    const bool spatial = part_spatial(c);
    std::vector<int32_t> perm(count);
    for (int i = 0; i < count; ++i) perm[i] = spatial ? c->sp_nodes[size_t(i)] : first + i; // ascending node index
    c->rx_sorted = false;
    if (is_geometric(c) && count > rm::kGroup) {
        std::vector<KdItem> items(static_cast<size_t>(count));
        for (int i = 0; i < count; ++i) {
            const int k = perm[size_t(i)];
            items[size_t(i)] = KdItem{{c->x[k], c->y[k], c->z[k]}, k, 0};
        }
        // Several channels in the table (BASELINE configs[3]: 16): receivers are ordered by CHANNEL first, by place inside a
        // channel -- a group of 64 then holds one channel (two, where two classes meet), the filter drops a frame for the
        // whole group by its channel mask (NodesDev::grp_chmask) before any distance is computed, and a frame's near
        // groups are its co-channel ones: a sixteenth of the pair tests.  The order is engine-internal: results are ranked by
        // node index whatever it is.  (RM_CHANNEL_ORDER=0: by place only.)
        std::vector<std::pair<int, int>> classes; // [lo, hi) of every channel class
        {
            static const bool off = [] {
                const char *e = std::getenv("RM_CHANNEL_ORDER");
                return e && std::atoi(e) == 0;
            }();
            bool several = false;
            for (int i = 1; i < count && !several; ++i) several = c->channel[size_t(items[size_t(i)].idx)] != c->channel[size_t(items[0].idx)];
            if (several && !off) {
                std::stable_sort(items.begin(), items.end(), [c](const KdItem &a, const KdItem &b) { return c->channel[size_t(a.idx)] < c->channel[size_t(b.idx)]; });
                int lo = 0;
                for (int i = 1; i <= count; ++i)
                    if (i == count || c->channel[size_t(items[size_t(i)].idx)] != c->channel[size_t(items[size_t(lo)].idx)]) {
                        classes.push_back({lo, i});
                        lo = i;
                    }
                if (classes.size() > 64 || size_t(count) / classes.size() < 4 * size_t(rm::kGroup)) classes.clear(); // (too many, too small: by place after all)
            }
        }
        if (classes.empty()) {
            kd_split(items.data(), 0, count, 3); // up to 8 threads
        } else {
            for (const auto &cl : classes) {
                // the class begins in the middle of a group of 64: its nearest receivers (along its widest axis) fill that group up,
                // the rest is split from a group boundary on
                int lo = cl.first;
                const int hi = cl.second;
                const int head = (rm::kGroup - lo % rm::kGroup) % rm::kGroup;
                if (head > 0 && hi - lo > head) {
                    double mn[3], mx[3];
                    for (int a = 0; a < 3; ++a) mn[a] = mx[a] = items[size_t(lo)].v[a];
                    for (int i = lo + 1; i < hi; ++i)
                        for (int a = 0; a < 3; ++a) {
                            mn[a] = std::min(mn[a], items[size_t(i)].v[a]);
                            mx[a] = std::max(mx[a], items[size_t(i)].v[a]);
                        }
                    int axis = 0;
                    for (int a = 1; a < 3; ++a)
                        if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
                    std::nth_element(items.begin() + lo, items.begin() + lo + head, items.begin() + hi, [axis](const KdItem &a, const KdItem &b) {
                        return a.v[axis] < b.v[axis] || (a.v[axis] == b.v[axis] && a.idx < b.idx);
                    });
                    lo += head;
                }
                kd_split(items.data(), lo, hi, 3);
            }
        }
        for (int i = 0; i < count; ++i) perm[size_t(i)] = items[size_t(i)].idx;
        c->rx_sorted = true;
    }
    std::vector<double> td;
    std::vector<int32_t> ti;
    std::vector<uint8_t> tb;
    RM_TRY(upload_gather(c->d_rx_x, c->x, perm, c->stream, td));
    RM_TRY(upload_gather(c->d_rx_y, c->y, perm, c->stream, td));
    RM_TRY(upload_gather(c->d_rx_z, c->z, perm, c->stream, td));
    RM_TRY(upload_gather(c->d_rx_rxprob, c->rxprob, perm, c->stream, td));
    RM_TRY(upload_gather(c->d_rx_channel, c->channel, perm, c->stream, ti));
    RM_TRY(upload_gather(c->d_rx_int_id, c->int_id, perm, c->stream, ti));
    RM_TRY(upload_gather(c->d_rx_enabled, c->enabled, perm, c->stream, tb));
    RM_HIP(c->d_rx_orig.ensure(std::max(count, 1)));
    RM_HIP(c->d_pos_of.ensure(std::max(span, 1)));
    std::vector<int32_t> pos_of(size_t(span), -1); // (a spatial partition: every node index, -1 = another rank's receiver)
    for (int i = 0; i < count; ++i) pos_of[size_t(perm[i] - first)] = i;
    if (count) RM_HIP(hipMemcpyAsync(c->d_rx_orig.p, perm.data(), size_t(count) * 4, hipMemcpyHostToDevice, c->stream));
    if (span) RM_HIP(hipMemcpyAsync(c->d_pos_of.p, pos_of.data(), size_t(span) * 4, hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    {
        std::vector<rm::RxRecord> recs(count);
        for (int i = 0; i < count; ++i) {
            const int k = perm[i];
            rm::RxRecord &r = recs[i];
            std::memset(&r, 0, sizeof(r));
            r.x = c->x[k];
            r.y = c->y[k];
            r.z = c->z[k];
            r.rxprob = c->rxprob[k];
            r.orig = k;
            r.int_id = c->int_id[k];
            r.channel = c->channel[k];
            r.enabled = c->enabled[k];
        }
        std::vector<rm::RxCompact> small(count);
        for (int i = 0; i < count; ++i) {
            const int k = perm[i];
            rm::RxCompact &r = small[i];
            r.x = c->x[k];
            r.y = c->y[k];
            r.z = c->z[k];
            r.orig = k;
            r.flags = (c->rxprob[k] != 1.0) ? 1u : 0u;
        }
        RM_HIP(c->d_rx_rec32.ensure(std::max(count, 1)));
        if (count)
            RM_HIP(hipMemcpyAsync(c->d_rx_rec32.p, small.data(), size_t(count) * sizeof(rm::RxCompact), hipMemcpyHostToDevice, c->stream));
        RM_HIP(c->d_rx_rec.ensure(std::max(count, 1)));
        if (count) {
            RM_HIP(hipMemcpyAsync(c->d_rx_rec.p, recs.data(), size_t(count) * sizeof(rm::RxRecord), hipMemcpyHostToDevice, c->stream));
            RM_HIP(hipStreamSynchronize(c->stream));
        }
    }
    c->h_pos_of.swap(pos_of);
    const int groups = (count + rm::kGroup - 1) / rm::kGroup;
    c->g_box.assign(size_t(groups), rm_context::GroupBox{});
    for (int g = 0; g < groups; ++g) {
        rm_context::GroupBox &b = c->g_box[size_t(g)];
        for (int i = g * rm::kGroup; i < std::min(count, (g + 1) * rm::kGroup); ++i) {
            const int k = perm[i];
            const double v[3] = {c->x[k], c->y[k], c->z[k]};
            for (int a = 0; a < 3; ++a) {
                if (i == g * rm::kGroup || v[a] < b.lo[a]) b.lo[a] = v[a];
                if (i == g * rm::kGroup || v[a] > b.hi[a]) b.hi[a] = v[a];
            }
        }
    }
    c->g_escaped.assign(size_t(groups), 0);
    c->table_sorts++;
    c->escaped_groups = 0;
    c->drifted_groups = 0;
    c->n_rx = count;
    c->rx_dirty = false;
    c->prefilter_dirty = true;
    return RM_OK;
}

// Changed nodes (host mirror already updated) go to the device in place: one launch, no
// synchronisation for a single node.  The receiver table keeps its engine order -- any permutation
// is correct, the order only decides how tight the groups' boxes are -- and is sorted again once
// enough receivers have left the box their group had when it was sorted.
int patch_nodes(rm_context *c, const int32_t *nodes, int count)
{
    if (count <= 0) return RM_OK;
    const int first = part_first(c), span = pos_span(c);
    static const long resort_after = [] {
        const char *e = std::getenv("RM_RESORT_AFTER"); // escaped groups that trigger a new sort (0: every change)
        return e ? std::atol(e) : -1L;
    }();
    const int groups = int(c->g_box.size());
    const long limit = resort_after >= 0 ? resort_after : std::max(2, groups / 128);
    std::vector<rm::NodePatch> list(static_cast<size_t>(count));
    bool frame_changed = false;
    for (int k = 0; k < count; ++k) {
        const int i = nodes[k];
        rm::NodePatch &p = list[size_t(k)];
        p.node = i;
        p.pos = -1;
        p.x = c->x[i]; p.y = c->y[i]; p.z = c->z[i];
        p.txpower = c->txpower[i]; p.txprob = c->txprob[i]; p.rxprob = c->rxprob[i];
        p.channel = c->channel[i];
        p.enabled = c->enabled[i];
        const double dv[3] = {p.x - c->org[0], p.y - c->org[1], p.z - c->org[2]};
        if (std::fabs(dv[0]) > c->coord_bound || std::fabs(dv[1]) > c->coord_bound || std::fabs(dv[2]) > c->coord_bound)
            frame_changed = true;
        if (c->rx_dirty || i < first || i >= first + span) continue;
        p.pos = c->h_pos_of[size_t(i - first)];
        if (!c->rx_sorted || p.pos < 0) continue;
        const int g = p.pos / rm::kGroup;
        const rm_context::GroupBox &b = c->g_box[size_t(g)];
        const double ext = std::max(b.hi[0] - b.lo[0], std::max(b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]));
        const double v[3] = {p.x, p.y, p.z};
        bool near = false, far = false;
        for (int a = 0; a < 3; ++a) {
            near = near || v[a] < b.lo[a] - 0.125 * ext || v[a] > b.hi[a] + 0.125 * ext;
            far = far || v[a] < b.lo[a] - 0.5 * ext || v[a] > b.hi[a] + 0.5 * ext;
        }
        uint8_t &flag = c->g_escaped[size_t(g)];
        if (near && !(flag & 1)) {
            flag |= 1;
            c->drifted_groups++;
        }
        if (far && !(flag & 2)) {
            flag |= 2;
            c->escaped_groups++;
        }
    }
    const rm::NodesDev nd = nodes_dev(c);
    if (count == 1) {
        RM_HIP(rm::launch_patch_nodes(c->stream, nd, nullptr, 1, list[0]));
    } else {
        RM_HIP(c->d_patch.ensure(size_t(count)));
        RM_HIP(hipMemcpyAsync(c->d_patch.p, list.data(), size_t(count) * sizeof(rm::NodePatch), hipMemcpyHostToDevice, c->stream));
        RM_HIP(rm::launch_patch_nodes(c->stream, nd, c->d_patch.p, count, list[0]));
        RM_HIP(hipStreamSynchronize(c->stream)); // the host list goes away
    }
    if (frame_changed) recompute_frame(c);
    c->prefilter_dirty = true;
    // a far-flung receiver makes its group a candidate for many frames; many slightly grown boxes cost as much
    if (c->rx_sorted && !c->rx_dirty && (c->escaped_groups > limit || (resort_after < 0 && c->drifted_groups > groups / 4)))
        c->rx_dirty = true; // sorted again before the next tick
    return RM_OK;
}

// ---- the node table's digest (rm_table_digest; rm_context::table_digest)
static uint64_t bits_of(double v)
{
    uint64_t b;
    std::memcpy(&b, &v, sizeof(b));
    return b;
}
static uint64_t node_hash(const rm_context *c, int i)
{
    uint64_t h = rm::host_mix64(0x9E3779B97F4A7C15ull * uint64_t(i + 1));
    const uint64_t f[8] = {bits_of(c->x[i]), bits_of(c->y[i]), bits_of(c->z[i]), bits_of(c->txpower[i]), bits_of(c->txprob[i]), bits_of(c->rxprob[i]),
                           (uint64_t(uint32_t(c->channel[i])) << 32) | uint64_t(uint32_t(c->int_id[i])), uint64_t(c->enabled[i] ? 1 : 0)};
    for (uint64_t v : f) h = rm::host_mix64(h ^ v);
    return h;
}
static void finish_digest(rm_context *c) { c->table_digest = rm::host_mix64(c->table_xor ^ rm::host_mix64(uint64_t(c->n) + 0xD1B54A32D192ED03ull)); }
static void digest_all(rm_context *c)
{
    c->table_xor = 0;
    for (int i = 0; i < c->n; ++i) c->table_xor ^= node_hash(c, i);
    finish_digest(c);
}

int prepare_nodes(rm_context *c)
{
    if (c->rx_dirty || c->prefilter_dirty) c->air.valid = false; // whatever the SINR lists' entries were computed from has changed
    if (c->rx_dirty) RM_TRY(rebuild_receivers(c));
    if (!c->prefilter_dirty) return RM_OK;
    const int groups = (c->n_rx + rm::kGroup - 1) / rm::kGroup;
    RM_HIP(c->d_rxf.ensure(std::max(c->n_rx, 1)));
    RM_HIP(c->d_bbox_xy.ensure(std::max(groups, 1)));
    RM_HIP(c->d_bbox_z.ensure(std::max(groups, 1)));
    RM_HIP(c->d_wg_box_xy.ensure(std::max(groups / 16 + 1, 1)));
    RM_HIP(c->d_grp_chmask.ensure(std::max(groups, 1)));
    RM_HIP(c->d_wg_chmask.ensure(std::max(groups / 16 + 1, 1)));
    RM_HIP(c->d_wg_box_z.ensure(std::max(groups / 16 + 1, 1)));
    RM_HIP(rm::launch_prep_rx(c->stream, nodes_dev(c), model_dev(c)));
    c->prefilter_dirty = false;
    return RM_OK;
}

} // namespace rmh

extern "C" {

int rm_nodes_upload(rm_context *c, int32_t n, const double *x, const double *y, const double *z,
                    const double *txpower, const int32_t *channel, const uint8_t *enabled, const double *rxprob,
                    const double *txprob, const int32_t *int_id)
{
    if (!c || n < 0) return fail(RM_ERR_INVALID, "bad arguments");
    if (n > 0 && (!x || !y)) return fail(RM_ERR_INVALID, "x and y are required");
    for (int i = 0; i < n; ++i) {
        if (!std::isfinite(x[i]) || !std::isfinite(y[i]) || (z && !std::isfinite(z[i])))
            return fail(RM_ERR_INVALID, "node positions must be finite");
    }
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(ev_flush_append(c)); // (an append left for the next drain reads the tick's records and the radio-state arrays as they are now)
    c->n = n;
    c->x.assign(x, x + n);
    c->y.assign(y, y + n);
    if (z) c->z.assign(z, z + n); else c->z.assign(n, 0.0);                          // Position.java:44-46
    if (txpower) c->txpower.assign(txpower, txpower + n); else c->txpower.assign(n, 0.0);   // Transciever.java:11
    if (channel) c->channel.assign(channel, channel + n); else c->channel.assign(n, 26);    // :12
    if (enabled) c->enabled.assign(enabled, enabled + n); else c->enabled.assign(n, 1);     // :13
    if (rxprob) c->rxprob.assign(rxprob, rxprob + n); else c->rxprob.assign(n, 1.0);        // :17
    if (txprob) c->txprob.assign(txprob, txprob + n); else c->txprob.assign(n, 1.0);        // :18
    if (int_id) c->int_id.assign(int_id, int_id + n);
    else {
        c->int_id.resize(n);
        for (int i = 0; i < n; ++i) c->int_id[i] = i + 1;
    }
    RM_TRY(upload(c->d_x, c->x, c->stream));
    RM_TRY(upload(c->d_y, c->y, c->stream));
    RM_TRY(upload(c->d_z, c->z, c->stream));
    RM_TRY(upload(c->d_txpower, c->txpower, c->stream));
    RM_TRY(upload(c->d_txprob, c->txprob, c->stream));
    RM_TRY(upload(c->d_rxprob_node, c->rxprob, c->stream));
    RM_TRY(upload(c->d_channel, c->channel, c->stream));
    RM_TRY(upload(c->d_int_id, c->int_id, c->stream));
    RM_TRY(upload(c->d_enabled, c->enabled, c->stream));
    {
        std::vector<rm::SrcRecord> sr(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) {
            rm::SrcRecord &r = sr[size_t(i)];
            r.x = c->x[i], r.y = c->y[i], r.z = c->z[i];
            r.txpower = c->txpower[i];
            r.txprob = c->txprob[i];
            r.channel = c->channel[i];
            r.int_id = c->int_id[i];
            r.pad[0] = r.pad[1] = 0.0;
        }
        RM_TRY(upload(c->d_srec, sr, c->stream));
        RM_HIP(hipStreamSynchronize(c->stream)); // (the host vector goes away)
    }
    RM_HIP(hipStreamSynchronize(c->stream));
    if (c->ev.on) RM_TRY(ev_ensure_nodes(c));
    recompute_frame(c);
    c->rx_dirty = true;
    c->frac_probs = -1;
    c->air_batches.clear();
    c->air_head = c->air_tail = 0;
    c->air.valid = false;
    c->pending.clear();
    if (c->rx_count >= 0 && c->rx_first + c->rx_count > n) {
        c->rx_first = 0;
        c->rx_count = -1;
    }
    c->rx_dirty = true;
    digest_all(c);
    return select_region(c); // a spatial partition: the same region of the new table
}

// keep the cached "can a draw happen" answer across a node change where that is possible
static void note_probabilities(rm_context *c, double old_rx, double old_tx, double new_rx, double new_tx)
{
    if (c->frac_probs < 0) return;
    const bool was = frac(old_rx) || frac(old_tx), is = frac(new_rx) || frac(new_tx);
    if (is) c->frac_probs = 1;
    else if (was && c->frac_probs == 1) c->frac_probs = -1; // it may have been the only one: scan again
}

int rm_node_update(rm_context *c, int32_t i, double x, double y, double z, double txpower, int32_t channel,
                   uint8_t enabled, double rxprob, double txprob)
{
    if (!c || i < 0 || i >= c->n) return fail(RM_ERR_INVALID, "node index out of range");
    if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) return fail(RM_ERR_INVALID, "position must be finite");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(ev_flush_append(c));
    note_probabilities(c, c->rxprob[i], c->txprob[i], rxprob, txprob);
    c->table_xor ^= node_hash(c, i);
    c->x[i] = x; c->y[i] = y; c->z[i] = z; c->txpower[i] = txpower; c->channel[i] = channel;
    c->enabled[i] = enabled; c->rxprob[i] = rxprob; c->txprob[i] = txprob;
    c->table_xor ^= node_hash(c, i);
    finish_digest(c);
    return patch_nodes(c, &i, 1);
}

int rm_nodes_move(rm_context *c, int32_t count, const int32_t *nodes, const double *x, const double *y, const double *z)
{
    if (!c || count < 0 || (count > 0 && (!nodes || !x || !y))) return fail(RM_ERR_INVALID, "bad arguments");
    for (int k = 0; k < count; ++k) {
        if (nodes[k] < 0 || nodes[k] >= c->n) return fail(RM_ERR_INVALID, "node index out of range");
        if (!std::isfinite(x[k]) || !std::isfinite(y[k]) || (z && !std::isfinite(z[k])))
            return fail(RM_ERR_INVALID, "position must be finite");
    }
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(ev_flush_append(c));
    for (int k = 0; k < count; ++k) {
        const int i = nodes[k];
        c->table_xor ^= node_hash(c, i);
        c->x[i] = x[k];
        c->y[i] = y[k];
        c->z[i] = z ? z[k] : 0.0; // Position.java:44-46: set(x, y) puts z at 0
        c->table_xor ^= node_hash(c, i);
    }
    finish_digest(c);
    return patch_nodes(c, nodes, count);
}

int rm_table_digest(const rm_context *c, uint64_t *digest)
{
    if (!c || !digest) return fail(RM_ERR_INVALID, "NULL argument");
    *digest = c->table_digest;
    return RM_OK;
}

int64_t rm_receiver_table_builds(const rm_context *c) { return c ? c->table_sorts : 0; }

int rm_node_count(const rm_context *c) { return c ? c->n : fail(RM_ERR_INVALID, "ctx is NULL"); }

int rm_set_partition(rm_context *c, int32_t first, int32_t count)
{
    if (!c || first < 0 || count < 0 || first + count > c->n) return fail(RM_ERR_INVALID, "partition out of range");
    RM_TRY(ev_flush_append(c)); // (whose Transcievers live here is part of what an append writes)
    if (c->air_culled && c->air_tail > c->air_head)
        return fail(RM_ERR_STATE, "frames on the air were kept for the present partition's region only: change the partition once they have left the air");
    c->rx_first = first;
    c->rx_count = count;
    c->sp_part = c->sp_parts = 0;
    c->sp_nodes.clear();
    c->rx_dirty = true;
    return RM_OK;
}

int rm_set_partition_spatial(rm_context *c, int32_t part, int32_t n_parts)
{
    if (!c || n_parts < 1 || part < 0 || part >= n_parts) return fail(RM_ERR_INVALID, "partition out of range");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(ev_flush_append(c));
    if (c->air_culled && c->air_tail > c->air_head && !(c->sp_parts == ((n_parts > 1) ? n_parts : 0) && c->sp_part == ((n_parts > 1) ? part : 0)))
        return fail(RM_ERR_STATE, "frames on the air were kept for the present partition's region only: change the partition once they have left the air");
    c->rx_first = 0;
    c->rx_count = -1;
    c->sp_part = (n_parts > 1) ? part : 0;
    c->sp_parts = (n_parts > 1) ? n_parts : 0;
    c->rx_dirty = true;
    return select_region(c);
}

int rm_partition_of_nodes(rm_context *c, int32_t n_parts, int32_t *part_of)
{
    if (!c || n_parts < 1 || (c->n > 0 && !part_of)) return fail(RM_ERR_INVALID, "bad arguments");
    std::vector<KdItem> items = all_items(c);
    region_split(items.data(), 0, c->n, 0, n_parts, -1, nullptr, part_of);
    return RM_OK;
}

int rm_region_split(int32_t n, const double *x, const double *y, const double *z, int32_t n_parts, int32_t *part_of)
{
    if (n < 0 || n_parts < 1 || (n > 0 && (!x || !y || !part_of))) return fail(RM_ERR_INVALID, "bad arguments");
    std::vector<KdItem> items(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) items[size_t(i)] = KdItem{{x[i], y[i], z ? z[i] : 0.0}, i, 0};
    region_split(items.data(), 0, n, 0, n_parts, -1, nullptr, part_of);
    return RM_OK;
}

int rm_partition_nodes(rm_context *c, int32_t *nodes, int32_t cap, int32_t *count)
{
    if (!c || !count || (cap > 0 && !nodes)) return fail(RM_ERR_INVALID, "bad arguments");
    const int k = part_count(c), first = part_first(c);
    *count = k;
    for (int i = 0; i < std::min(k, cap); ++i) nodes[i] = part_spatial(c) ? c->sp_nodes[size_t(i)] : first + i;
    return k > cap ? fail(RM_ERR_CAPACITY, "caller buffer too small for the partition's nodes") : RM_OK;
}

} // extern "C"
