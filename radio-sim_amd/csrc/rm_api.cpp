// rm_api.cpp -- C-ABI (include/radiomedium_hip.h) of the MI355X radio-medium engine: context,
// device-resident node state, on-air list, the per-tick launch sequence.  Compiled by hipcc.
//
// There is deliberately no CPU fallback in this file: every evaluation goes through the gfx950
// kernels of rm_filter / rm_exact / rm_reorder / rm_transmit .hip, and rm_create fails when no HIP device can be used.
//
// Reference paths: /root/reference/radio-medium/java/se/sics/emul8/radiomedium/.

#include "rm_engine.h"
#include "rm_evorder.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <memory>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define RM_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(RM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                \
    } while (0)

#define RM_TRY(call)                                                                                   \
    do {                                                                                               \
        int r_ = (call);                                                                               \
        if (r_ != RM_OK) return r_;                                                                    \
    } while (0)

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t want, bool keep = false, hipStream_t s = nullptr)
    {
        if (want <= n) return hipSuccess;
        size_t grow = std::max(want, n + n / 2);
        T *q = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&q), grow * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep && p && n) {
            e = hipMemcpyAsync(q, p, n * sizeof(T), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                (void)hipFree(q);
                return e;
            }
        }
        if (p) (void)hipFree(p);
        p = q;
        n = grow;
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

const char *model_name(int kind)
{
    switch (kind) {
    case RM_MODEL_NULL: return "Null radio medium - just forwards incoming packets to all other nodes"; // NullRadioMedium.java:44
    case RM_MODEL_UDGM: return "UDGM Radio Medium";                                                   // UDGMRadioMedium.java:28
    case RM_MODEL_UDGM_CONST: return "UDGM Constant Loss Radio Medium";                               // UDGMConstantLossRadioMedium.java:12
    case RM_MODEL_N2N: return "Matrix Radio Medium";                                                  // N2NRadioMedium.java:17
    case RM_MODEL_LOGDIST: return "Log-distance SINR Radio Medium (MI355X)";
    default: return "?";
    }
}

} // namespace

// Everything one evaluated tick owns on the device.  A context is its own slot 0; rm_batch_*
// adds further slots so that several ticks can be in flight through one launch sequence.
struct TickSlot {
    DevBuf<rm_tx_record> d_tx;   // records uploaded by the host / built from source indices
    DevBuf<float4> d_p_txf;      // per-frame pre-filter records
    DevBuf<int32_t> d_p_ch, d_p_src;
    DevBuf<float> d_p_inv;

    DevBuf<uint32_t> d_cnt, d_off, d_slot_tot, d_slot_off;
    DevBuf<uint32_t> d_counters; // two parities x 8: [1] dropped flag, [2..5] out_count
    DevBuf<uint32_t> d_shards;   // two parities x kShards x kShardStride append counters
    DevBuf<uint32_t> d_cursor, d_cand_tot, d_seg_off;
    DevBuf<int32_t> d_a_e;
    int zero_len = 0;        // slots of cursor / cand_tot that may be non-zero
    int parity = 0;
    DevBuf<int32_t> d_st_pkt, d_st_dst, d_st_next, d_head;
    DevBuf<uint32_t> d_st_blk;
    DevBuf<double> d_st_aux, d_st_lin, d_st_sinr, d_st_prob;
    DevBuf<int32_t> d_st_orig;
    DevBuf<uint8_t> d_st_flags, d_st_coll;
    DevBuf<int32_t> d_out_pkt, d_out_dst, d_a_pkt, d_a_dst;
    DevBuf<uint8_t> d_out_verdict, d_pkt_interf, d_a_verdict;
    DevBuf<double> d_out_rssi, d_out_sinr, d_out_prob, d_a_rssi, d_a_sinr, d_a_prob;
    DevBuf<uint32_t> d_draw_scan, d_scan_block;
    DevBuf<uint64_t> d_pkt_rng;
    DevBuf<uint32_t> d_pkt_draw_cnt, d_all_cnt;
    bool draws_pending = false; // partitioned + probabilistic: waiting for rm_tick_finish_draws
    rm::ModelDev pending_model{};
    uint32_t alloc_cap = 0;
    int alloc_feat = 0; // kFeat* buffers allocated at alloc_cap

    // last tick
    rm::TickDev last{};
    int last_n_new = 0;
    bool have_result = false;
    int64_t last_links = 0;
    // the closed-loop tick (rm_tick.hip) leaves per-frame ordered segments; the compact packet-major
    // arrays of rm_device_result are produced (k_reorder) when somebody asks for them
    bool compact_pending = false;
    rm::ModelDev last_model{};
    rm::LaunchCfg last_cfg{};

    void release_all();
};

struct rm_context : TickSlot {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    rm_model_params params{};
    double base_rssi = -100.0; // AbstractRadioMedium.java:38

    // host mirror of the node table (Simulator.getNodes() snapshot)
    int n = 0;
    std::vector<double> x, y, z, txpower, rxprob, txprob;
    std::vector<int32_t> channel, int_id;
    std::vector<uint8_t> enabled;
    // device-resident source table (SoA, node-index order): what a packet copies from its source
    DevBuf<double> d_x, d_y, d_z, d_txpower, d_txprob;
    DevBuf<int32_t> d_channel, d_int_id;
    // device-resident receiver table of this partition (SoA, engine order = spatially sorted)
    DevBuf<double> d_rx_x, d_rx_y, d_rx_z, d_rx_rxprob;
    DevBuf<int32_t> d_rx_channel, d_rx_int_id, d_rx_orig, d_pos_of;
    DevBuf<uint8_t> d_rx_enabled;
    DevBuf<rm::RxRecord> d_rx_rec;
    DevBuf<rm::RxCompact> d_rx_rec32;
    DevBuf<float4> d_rxf, d_bbox_xy, d_wg_box_xy;
    DevBuf<float2> d_wg_box_z;
    DevBuf<float2> d_bbox_z;
    DevBuf<double> d_n2n;
    int n2n_m = 0;
    DevBuf<uint32_t> d_shadow_tbl;
    bool shadow_tbl_valid = false;
    int n_rx = 0;            // receivers in the table
    bool rx_sorted = false;  // engine order != node-index order
    // changed nodes are written in place while the engine order is still a good spatial order
    struct GroupBox {
        double lo[3], hi[3];
    };
    std::vector<int32_t> h_pos_of;  // node index - rx_first -> engine position (host copy of d_pos_of)
    std::vector<GroupBox> g_box;    // per group of 64: its box when the table was sorted
    std::vector<uint8_t> g_escaped; // bit 0 / 1: a receiver of this group has left the box by more than 1/8 / 1/2 of its extent
    int drifted_groups = 0, escaped_groups = 0;
    int64_t table_sorts = 0;        // times the receiver table was (re)built
    DevBuf<rm::NodePatch> d_patch;

    // reception stage (rm_events.hip): pending packets + their links, radio state per node, delivery list block
    struct Events {
        bool on = false;
        uint32_t pk_cap = 0, pool_cap = 0, g_cap = 0;
        DevBuf<rm::EvState> d_st;
        DevBuf<rm::EvPacket> d_pk;
        DevBuf<int32_t> d_ldst;
        DevBuf<double> d_lrssi;
        DevBuf<uint8_t> d_lverdict;
        DevBuf<int64_t> d_gtime;
        DevBuf<uint64_t> d_gmeta;
        DevBuf<uint32_t> d_gref, d_grank, d_cnt, d_off;
        DevBuf<unsigned long long> d_recv_key, d_send_key;
        DevBuf<uint8_t> d_receiving, d_sending;
        DevBuf<double> d_latched;
        int state_n = 0;          // nodes the radio-state arrays hold
        char *h_out = nullptr;    // host-mapped: EvHeader + packet / dst / rssi arrays of pool_cap entries
        char *h_info = nullptr;   // host-mapped: node-info answers
        int info_n = 0;
        uint32_t seq = 0, info_seq = 0;
        DevBuf<int32_t> d_info_nodes;
        int64_t next_packet = 0;  // host mirror of EvTails::gseq_next
        int par = 0;              // which EvState::tails are current (flips with every appended tick)
    } ev;
    DevBuf<uint8_t> d_enabled;   // Transciever.isEnabled by node index

    int frac_probs = -1;         // cached: any rx/tx probability strictly between 0 and 1 (-1 = unknown)
    bool tick_frac_records = false; // a host record of the running tick has 0 < txprob < 1 (rm_enqueue_tx_records)
    bool rx_dirty = true;        // receiver table has to be rebuilt (positions / partition / model class)
    bool prefilter_dirty = true; // pre-filter records have to be recomputed
    double org[3] = {0, 0, 0};
    double coord_bound = 0, f32_slack = 0;

    int rx_first = 0, rx_count = -1; // -1 = all nodes
    uint32_t cap = 1u << 22;

    int64_t current_time = 0;
    int64_t t_begin = 0, t_end = 0;
    bool in_tick = false;

    // on-air list (host-record mode)
    std::vector<rm_tx_record> onair;   // frames of earlier ticks still on the air (SINR mode)
    std::vector<rm_tx_record> pending; // frames enqueued in the current tick
    // on-air list (device-source mode, SINR): the live batches are a window [air_head, air_tail) of
    // d_air; a batch = the frames of one rm_tick_run_sources_device call (same start and air time)
    struct AirBatch {
        int count;
        int64_t end_us;
        uint32_t tick; // AirLists::tick of the call that put the batch on the air
    };
    DevBuf<rm_tx_record> d_air;
    std::vector<AirBatch> air_batches;
    size_t air_head = 0, air_tail = 0;
    // the per-receiver interferer lists of the frames on the air, alive on the device from tick to tick (rm::AirDev):
    // a SINR tick evaluates its new frames only, as long as nothing the old entries were computed from has changed
    struct AirLists {
        DevBuf<rm::AirEntry> pool;
        DevBuf<unsigned long long> head;
        DevBuf<uint32_t> tail, mark, bad;
        bool valid = false;       // the lists hold exactly the frames on the air
        uint32_t tick = 0;        // number of the last tick that added entries (1 ..)
        uint32_t sub_cap = 0;     // entries per sub-ring (a power of two)
        int64_t last_t_begin = 0;
        uint64_t rebuilds = 0, incremental = 0;
    } air;
    std::vector<uint32_t> onair_tick; // AirLists::tick per frame of `onair`
    bool dev_records_from_caller = false; // the tick being prepared takes rm_tx_record arrays the caller built in device memory
    mutable rm::ModelDev mdev{};            // model_dev()'s last answer and what it was derived from
    mutable unsigned char mdev_key[320] = {};
    mutable bool mdev_valid = false;

    DevBuf<uint64_t> d_rng;  // [1] java.util.Random state, shared by all slots
    rm::TransmitResult *h_transmit = nullptr; // host-mapped result block of rm_transmit
    // host-mapped result block of rm_tick_flush*: the last kernel of a flushed tick writes header,
    // offsets and records there, the host waits for the header's sequence number
    char *h_stage = nullptr;
    uint32_t stage_links = 0, stage_packets = 0, stage_seq = 0;
    DevBuf<uint32_t> d_pack_done;
    DevBuf<rm::PackSlot> d_pack;    // descriptors of rm_batch_result_view
    rm::PackSlot *h_pack = nullptr; // their pinned staging
    // pinned staging of the Tx records of rm_tick_begin / rm_enqueue_tx* (two buffers, each guarded by an event)
    rm_tx_record *h_tx[2] = {nullptr, nullptr};
    size_t h_tx_n[2] = {0, 0};
    hipEvent_t h_tx_ev[2] = {nullptr, nullptr};
    int h_tx_gen = 0;
    uint32_t transmit_seq = 0;
    DevBuf<rm::TickDev> d_ticks; // [RM_MAX_BATCH] descriptors of the running rm_batch_* call
    // larger batches: k_fetch_ticks reads them from pinned, host-mapped memory (two staging buffers, each
    // guarded by an event: it is rewritten only after the kernel that read it has completed)
    rm::TickDev *h_ticks[2] = {nullptr, nullptr};
    hipEvent_t h_ticks_ev[2] = {nullptr, nullptr};
    int h_ticks_gen = 0;
    std::vector<std::unique_ptr<TickSlot>> extra_slots; // result slots 1.. of rm_batch_*

    // instantiated hipGraphs of the per-tick launch sequence, keyed by a hash of every launch argument
    struct GraphEntry {
        uint64_t key;
        hipGraphExec_t exec;
        uint64_t last_use;
    };
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0;
    bool use_graphs = false; // RM_GRAPH=1: replay the tick from a cached hipGraph (measured slower than eager
                             // launches on ROCm 7.2 for this 5-kernel sequence: 50 vs 46 us per tick)

    // profiling of the dominant kernel
    bool profile = false;   // sampling on
    int profile_every = 1;  // take an event-timed sample every n-th tick
    uint64_t tick_index = 0;
    // a sampled tick records one event before every stage and one after the last
    struct Sample {
        hipEvent_t ev[RM_PROFILE_STAGES + 1];
        int stage[RM_PROFILE_STAGES];
        int n = 0;
    };
    std::vector<Sample> ev_pool;
    size_t ev_used = 0;
    uint32_t prof_samples = 0;
    double prof_ms[RM_PROFILE_STAGES] = {0};
};

void TickSlot::release_all()
{
    d_tx.release(); d_p_txf.release(); d_p_ch.release(); d_p_src.release(); d_p_inv.release();
    d_cnt.release(); d_off.release(); d_slot_tot.release(); d_slot_off.release();
    d_counters.release(); d_shards.release(); d_cursor.release(); d_cand_tot.release(); d_seg_off.release(); d_a_e.release();
    d_st_pkt.release(); d_st_dst.release(); d_st_next.release(); d_head.release(); d_st_blk.release(); d_st_aux.release();
    d_st_lin.release(); d_st_sinr.release(); d_st_prob.release(); d_st_orig.release(); d_st_flags.release(); d_st_coll.release();
    d_out_pkt.release(); d_out_dst.release(); d_a_pkt.release(); d_a_dst.release(); d_out_verdict.release(); d_pkt_interf.release();
    d_a_verdict.release(); d_out_rssi.release(); d_out_sinr.release(); d_out_prob.release(); d_a_rssi.release(); d_a_sinr.release();
    d_a_prob.release(); d_draw_scan.release(); d_scan_block.release(); d_pkt_rng.release(); d_pkt_draw_cnt.release(); d_all_cnt.release();
    alloc_cap = 0;
    alloc_feat = 0;
    have_result = false;
}

namespace {

bool is_sinr(const rm_context *c) { return c->params.kind == RM_MODEL_LOGDIST && (c->params.flags & RM_LD_SINR); }

int part_first(const rm_context *c) { return c->rx_count < 0 ? 0 : c->rx_first; }
int part_count(const rm_context *c) { return c->rx_count < 0 ? c->n : c->rx_count; }

bool frac(double p) { return p > 0.0 && p < 1.0; }

// can a java.util.Random draw ever be consumed with the current model + node table?
// (the O(N) scan of the probabilities is cached until the node table changes)
bool maybe_draws(rm_context *c)
{
    const int k = c->params.kind;
    if (k == RM_MODEL_NULL || k == RM_MODEL_UDGM_CONST) return false;
    if (k == RM_MODEL_N2N) return true;
    if (c->tick_frac_records) return true; // a record of this tick carries its own fractional txProbability
    if (k == RM_MODEL_UDGM && c->params.udgm_success_ratio_rx != 1.0) return true;
    if (c->frac_probs < 0) {
        c->frac_probs = 0;
        for (int i = 0; i < c->n; ++i)
            if (frac(c->rxprob[i]) || frac(c->txprob[i])) {
                c->frac_probs = 1;
                break;
            }
    }
    return c->frac_probs == 1;
}

int validate_model(const rm_model_params *p)
{
    if (p->kind < RM_MODEL_NULL || p->kind > RM_MODEL_LOGDIST) return fail(RM_ERR_INVALID, "unknown model kind");
    if (p->kind == RM_MODEL_LOGDIST) {
        if (!(p->ld_d0 > 0.0) || !(p->ld_exponent >= 0.0) || !(p->ld_sigma_db >= 0.0) || !(p->ld_clip >= 0.0))
            return fail(RM_ERR_INVALID, "logdist: need d0 > 0, exponent >= 0, sigma >= 0, clip >= 0");
    }
    return RM_OK;
}

void recompute_frame(rm_context *c)
{
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int i = 0; i < c->n; ++i) {
        const double v[3] = {c->x[i], c->y[i], c->z[i]};
        for (int a = 0; a < 3; ++a) {
            if (i == 0 || v[a] < lo[a]) lo[a] = v[a];
            if (i == 0 || v[a] > hi[a]) hi[a] = v[a];
        }
    }
    double bound = 0;
    for (int a = 0; a < 3; ++a) {
        c->org[a] = 0.5 * (lo[a] + hi[a]);
        bound = std::max(bound, std::max(hi[a] - c->org[a], c->org[a] - lo[a]));
    }
    c->coord_bound = bound * (1.0 + 1e-9) + 1e-300;
    // fp32 frame: each coordinate is off by at most 2^-24 * bound; see DESIGN.md "Pre-filter"
    c->f32_slack = 4.0 * std::sqrt(3.0) * std::ldexp(1.0, -24) * c->coord_bound;
    c->prefilter_dirty = true;
}

rm::ModelDev model_dev(const rm_context *c)
{
    // planned once per tick of a batch (and more than once): the derived values (a software pow10, the mixed seed) are
    // kept as long as everything they come from is unchanged
    struct Key {
        rm_model_params p;
        double org[3], coord_bound, f32_slack;
        const void *n2n, *shadow;
        int n2n_m;
    };
    Key key;
    std::memset(&key, 0, sizeof(key)); // (padding bytes take part in the comparison)
    key.p = c->params;
    key.org[0] = c->org[0]; key.org[1] = c->org[1]; key.org[2] = c->org[2];
    key.coord_bound = c->coord_bound;
    key.f32_slack = c->f32_slack;
    key.n2n = c->d_n2n.p;
    key.shadow = c->shadow_tbl_valid ? c->d_shadow_tbl.p : nullptr;
    key.n2n_m = c->n2n_m;
    static_assert(sizeof(Key) <= sizeof(c->mdev_key), "model key buffer");
    if (c->mdev_valid && std::memcmp(&key, c->mdev_key, sizeof(key)) == 0) return c->mdev;
    rm::ModelDev m{};
    const rm_model_params &p = c->params;
    m.kind = p.kind;
    m.flags = p.flags;
    m.udgm_ratio_rx = p.udgm_success_ratio_rx;
    m.udgm_range = p.udgm_transmission_range;
    m.const_range = p.const_range;
    m.n2n = c->d_n2n.p;
    m.n2n_m = c->n2n_m;
    m.ld_pl0 = p.ld_pl0_db;
    m.ld_exp = p.ld_exponent;
    m.ld_d0 = p.ld_d0;
    m.ld_sigma = p.ld_sigma_db;
    m.ld_clip = p.ld_clip;
    m.ld_seed_mixed = rm::host_mix64(p.ld_seed + 0x9E3779B97F4A7C15ull);
    m.ld_sens = p.ld_sensitivity_dbm;
    m.ld_noise = p.ld_noise_dbm;
    m.ld_capture = p.ld_capture_db;
    m.ld_ifloor = p.ld_ifloor_dbm;
    m.ld_noise_lin = rm::host_det_pow10(p.ld_noise_dbm / 10.0);
    m.ld_level = p.ld_sensitivity_dbm;
    m.ld_cut_scale = (p.ld_exponent > 0.0) ? 3.3219280948873622 / (10.0 * p.ld_exponent) : 0.0;
    if ((p.flags & RM_LD_SINR) && p.ld_ifloor_dbm < m.ld_level) m.ld_level = p.ld_ifloor_dbm;
    m.org_x = c->org[0];
    m.org_y = c->org[1];
    m.org_z = c->org[2];
    m.coord_bound = c->coord_bound;
    m.f32_slack = c->f32_slack;
    m.geo_cut = -1.0;
    m.shadow_tbl = c->shadow_tbl_valid ? c->d_shadow_tbl.p : nullptr;
    if (p.kind == RM_MODEL_UDGM) {
        const double r = std::fabs(p.udgm_transmission_range);
        m.geo_cut = (r == 0.0) ? -1.0 : r * (1.0 + 1e-9); // ratio > 1.0 -> unheard, d == range is in
    } else if (p.kind == RM_MODEL_UDGM_CONST) {
        m.geo_cut = (p.const_range > 0.0) ? p.const_range * (1.0 + 1e-9) : -1.0; // strict distance < range
    }
    std::memcpy(c->mdev_key, &key, sizeof(key));
    c->mdev = m;
    c->mdev_valid = true;
    return m;
}

rm::NodesDev nodes_dev(rm_context *c)
{
    rm::NodesDev nd{};
    nd.n = c->n;
    nd.sx = c->d_x.p;
    nd.sy = c->d_y.p;
    nd.sz = c->d_z.p;
    nd.stxpower = c->d_txpower.p;
    nd.stxprob = c->d_txprob.p;
    nd.schannel = c->d_channel.p;
    nd.sint_id = c->d_int_id.p;
    nd.senabled = c->d_enabled.p;
    nd.n_rx = c->n_rx;
    nd.x = c->d_rx_x.p;
    nd.y = c->d_rx_y.p;
    nd.z = c->d_rx_z.p;
    nd.rxprob = c->d_rx_rxprob.p;
    nd.channel = c->d_rx_channel.p;
    nd.int_id = c->d_rx_int_id.p;
    nd.orig = c->d_rx_orig.p;
    nd.enabled = c->d_rx_enabled.p;
    nd.rec = c->d_rx_rec.p;
    static const bool no_rec32 = std::getenv("RM_NO_REC32") != nullptr;
    nd.rec32 = no_rec32 ? nullptr : c->d_rx_rec32.p;
    nd.pos_of = c->d_pos_of.p;
    nd.rx_first = part_first(c);
    nd.rxf = c->d_rxf.p;
    nd.bbox_xy = c->d_bbox_xy.p;
    nd.bbox_z = c->d_bbox_z.p;
    nd.wg_box_xy = c->d_wg_box_xy.p;
    nd.wg_box_z = c->d_wg_box_z.p;
    return nd;
}

bool is_geometric(const rm_context *c)
{
    const int k = c->params.kind;
    return k == RM_MODEL_UDGM || k == RM_MODEL_UDGM_CONST || k == RM_MODEL_LOGDIST;
}

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

// (re)build the receiver table of the partition in engine order
int rebuild_receivers(rm_context *c)
{
    const int first = part_first(c), count = part_count(c);
    std::vector<int32_t> perm(count);
    for (int i = 0; i < count; ++i) perm[i] = first + i;
    c->rx_sorted = false;
    if (is_geometric(c) && count > rm::kGroup) {
        std::vector<KdItem> items(static_cast<size_t>(count));
        for (int i = 0; i < count; ++i) {
            const int k = first + i;
            items[size_t(i)] = KdItem{{c->x[k], c->y[k], c->z[k]}, k, 0};
        }
        kd_split(items.data(), 0, count, 3); // up to 8 threads
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
    RM_HIP(c->d_pos_of.ensure(std::max(count, 1)));
    std::vector<int32_t> pos_of(count);
    for (int i = 0; i < count; ++i) pos_of[perm[i] - first] = i;
    if (count) {
        RM_HIP(hipMemcpyAsync(c->d_rx_orig.p, perm.data(), size_t(count) * 4, hipMemcpyHostToDevice, c->stream));
        RM_HIP(hipMemcpyAsync(c->d_pos_of.p, pos_of.data(), size_t(count) * 4, hipMemcpyHostToDevice, c->stream));
        RM_HIP(hipStreamSynchronize(c->stream));
    }
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
    const int first = part_first(c), pcount = part_count(c);
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
        if (c->rx_dirty || i < first || i >= first + pcount) continue;
        p.pos = c->h_pos_of[size_t(i - first)];
        if (!c->rx_sorted) continue;
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

// Link-sized buffers of a result slot.  Only what the configuration touches is allocated (a
// batch keeps up to RM_MAX_BATCH slots): `payload` = per-entry rssi / probability / node index
// (unsorted tables and SINR), `sinr` = per-receiver lists and linear powers, `draws` = the
// java.util.Random scan and the probabilities carried to it.
enum { kFeatPayload = 1, kFeatSinr = 2, kFeatDraws = 4 };
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

// Second-level filter of the shadowed log-distance medium: for a link at rho = d^2/cut^2 the
// deviate may be at most x(rho) = (5 n log10(1/rho) - sigma*clip)/sigma for the link to reach the
// candidate level, i.e. the hash's uniform at most Phi(x).  One conservative 32-bit threshold per
// bin (lower bin edge, rho padded by kShadowPad, x padded for the quantile approximation).
int build_shadow_table(rm_context *c)
{
    const rm_model_params &p = c->params;
    c->shadow_tbl_valid = false;
    if (p.kind != RM_MODEL_LOGDIST || !(p.ld_sigma_db > 0.0) || !(p.ld_exponent > 0.0)) return RM_OK;
    std::vector<uint32_t> tbl(rm::kShadowBins);
    for (int b = 0; b < rm::kShadowBins; ++b) {
        const double rho = (double(b) / rm::kShadowBins) * (1.0 - rm::kShadowPad);
        double umax = 1.0;
        if (rho > 0.0) {
            const double margin = 5.0 * p.ld_exponent * std::log10(1.0 / rho) - p.ld_sigma_db * p.ld_clip;
            const double x = margin / p.ld_sigma_db + 1e-6;
            if (x < p.ld_clip) umax = 0.5 * std::erfc(-x / std::sqrt(2.0));
        }
        const double v = std::floor(umax * 4294967296.0) + 2.0;
        tbl[b] = v >= 4294967295.0 ? 0xFFFFFFFFu : uint32_t(v);
    }
    RM_HIP(c->d_shadow_tbl.ensure(rm::kShadowBins));
    RM_HIP(hipMemcpyAsync(c->d_shadow_tbl.p, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    c->shadow_tbl_valid = true;
    return RM_OK;
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
    RM_HIP(c->d_wg_box_z.ensure(std::max(groups / 16 + 1, 1)));
    RM_HIP(rm::launch_prep_rx(c->stream, nodes_dev(c), model_dev(c)));
    c->prefilter_dirty = false;
    return RM_OK;
}

// What one tick's launch sequence needs besides the slot: filled by prepare_tick.
struct TickPlan {
    rm::TickDev t{};
    rm::LaunchCfg cfg{};
    bool sinr = false, stochastic = false, partitioned = false;
    bool empty = false; // nothing to sweep: the result is an empty one
};

// Buffers and descriptor of one tick in result slot `ts`; `tx` is the on-air list in device memory
// (build mode: where the records of the source indices `src_list` are written).
enum { kAirNone = 0, kAirIncremental = 1, kAirRebuild = 2 };

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

int prepare_tick(rm_context *c, TickSlot &ts, TickPlan &plan, bool want_wg, const rm_tx_record *tx, int n_active,
                 int first_new, const int32_t *src_list = nullptr, int64_t src_start_us = 0, int64_t src_air_us = 0,
                 int air_mode = kAirNone, uint32_t air_oldest = 0, const rm::PlanKnobs *knobs_in = nullptr)
{
    const rm::PlanKnobs knobs = knobs_in ? *knobs_in : rm::read_plan_knobs();
    const int n_new = n_active - first_new;
    ts.have_result = false;
    ts.compact_pending = false;
    ts.last_n_new = n_new;
    RM_TRY(prepare_nodes(c));

    const bool sinr = is_sinr(c);
    const bool stochastic = maybe_draws(c);
    RM_TRY(ensure_link_buffers(c, ts, ((!c->rx_sorted || sinr) ? kFeatPayload : 0) | (sinr ? kFeatSinr : 0) |
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
    t.check_txprob = (!stochastic && !src_list && c->dev_records_from_caller && c->params.kind != RM_MODEL_NULL && c->params.kind != RM_MODEL_UDGM_CONST) ? 1 : 0;
    t.first_eval = sinr ? 0 : first_new;
    const bool nothing_to_sweep = (n_new <= 0 || rx_count <= 0); // no launch at all: the lists stay as they are
    if (sinr && air_mode == kAirRebuild && nothing_to_sweep) c->air.valid = false; // rebuilt with the next frames
    if (sinr && air_mode == kAirIncremental && nothing_to_sweep) c->air.last_t_begin = c->t_begin;
    if (sinr && air_mode != kAirNone && !nothing_to_sweep) {
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
    t.out_sinr = sinr ? ts.d_out_sinr.p : nullptr; // only the SINR extension writes it: 8 of a record's 25 bytes
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
    // The launch sequence.  On a sampled tick (rm_profile_enable) every stage is bracketed by HIP
    // events on the stream; otherwise the stages are launched back to back (or, with RM_GRAPH=1,
    // replayed from an instantiated hipGraph keyed by the launch arguments).
    const bool sample = c->profile && (c->tick_index++ % uint64_t(c->profile_every) == 0);
    rm_context::Sample *smp = nullptr;
    if (sample) {
        if (c->ev_used == c->ev_pool.size()) {
            rm_context::Sample ns;
            for (auto &e : ns.ev) RM_HIP(hipEventCreate(&e));
            c->ev_pool.push_back(ns);
        }
        smp = &c->ev_pool[c->ev_used++];
        smp->n = 0;
    }
    auto stage = [&](int id) -> int {
        if (smp) {
            RM_HIP(hipEventRecord(smp->ev[smp->n], s));
            smp->stage[smp->n++] = id;
        }
        return RM_OK;
    };
    // frames per tick up to which a lone tick takes the one-frame-per-workgroup path (RM_FRAME_TICK=0: never)
    static const int frame_tick_max = [] {
        const char *e = std::getenv("RM_FRAME_TICK");
        return e ? std::atoi(e) : 4096;
    }();
    const int seg_len = (t.n_active - t.first_new <= frame_tick_max) ? rm::frame_tick_segment(t, cfg, m) : 0;
    auto sequence = [&]() -> int {
        const bool air = sinr && t.air.pool != nullptr;
        const bool air_in_prep = air && t.filter_mode == rm::kFilterWg; // k_tick_prep leaves the SELF entries and looks at the sticky flag
        if (air && !air_in_prep) RM_HIP(rm::launch_air_begin(s, t));
        else if (sinr && !air) RM_HIP(hipMemsetAsync(ts.d_head.p, 0xFF, size_t(rx_count) * sizeof(int32_t), s));
        if (smp) RM_TRY(stage(RM_STAGE_EMPTY)); // calibration: an empty bracket
        if (seg_len > 0) {
            // the closed-loop tick: filter, exact evaluation and node order of a frame inside one workgroup
            // (rm_tick.hip) -- ONE launch; the compact arrays only for the draw kernels, or on demand
            RM_TRY(stage(RM_STAGE_FILTER));
            RM_HIP(rm::launch_tick_frames(s, nd, m, t, cfg, seg_len));
            if (stochastic) {
                RM_TRY(stage(RM_STAGE_REORDER));
                rm::TickDev tr = t;
                tr.seg_ordered = 1;
                RM_HIP(rm::launch_reorder(s, m, tr, cfg));
                RM_TRY(stage(RM_STAGE_DRAWS));
                RM_HIP(rm::launch_draws_scan(s, t));
                if (!partitioned) RM_HIP(rm::launch_draws_apply(s, m, t, nullptr, 1, 0));
            }
            if (smp) RM_HIP(hipEventRecord(smp->ev[smp->n], s));
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
        }
        if (smp) RM_HIP(hipEventRecord(smp->ev[smp->n], s));
        return RM_OK;
    };
    if (c->use_graphs && !sample) {
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
    ts.compact_pending = seg_len > 0 && !stochastic;
    ts.last.seg_ordered = (seg_len > 0) ? 1 : 0;
    ts.last_model = m;
    ts.last_cfg = cfg;
    ts.have_result = true;
    return RM_OK;
}

// the compact packet-major arrays of a tick that so far only has its per-frame segments
int materialize(rm_context *c, TickSlot &ts)
{
    if (!ts.compact_pending) return RM_OK;
    RM_HIP(rm::launch_reorder(c->stream, ts.last_model, ts.last, ts.last_cfg));
    ts.compact_pending = false;
    return RM_OK;
}

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
    e.g_cap = v.g_cap;
    e.recv_key = v.d_recv_key.p;
    e.send_key = v.d_send_key.p;
    e.receiving = v.d_receiving.p;
    e.sending = v.d_sending.p;
    e.latched = v.d_latched.p;
    e.n_nodes = v.state_n;
    e.own_first = part_first(c);
    e.own_count = part_count(c);
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
int ev_append(rm_context *c, TickSlot &ts)
{
    if (!c->ev.on || !ts.have_result || ts.last_n_new <= 0) {
        return RM_OK;
    }
    RM_TRY(ev_ensure_nodes(c));
    const rm::TickDev &t = ts.last;
    rm::EvLinkSrc ls{};
    const uint32_t *dropped = nullptr;
    if (part_count(c) <= 0) { // no receivers here: the packets still count (and their transmission events, elsewhere)
        static_assert(sizeof(uint32_t) == 4, "");
    }
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
    const int immediate = (c->params.kind == RM_MODEL_UDGM_CONST) ? 1 : 0;
    RM_HIP(rm::launch_ev_append(c->stream, ev_dev(c), ls, t.tx + t.first_new, ts.last_n_new, c->current_time, immediate, dropped));
    c->ev.par ^= 1; // the launch wrote the other set of tails
    c->ev.next_packet += ts.last_n_new;
    return RM_OK;
}

int run_tick(rm_context *c, const rm_tx_record *tx, int n_active, int first_new, const int32_t *src_list = nullptr,
             int64_t src_start_us = 0, int64_t src_air_us = 0, int air_mode = kAirNone, uint32_t air_oldest = 0)
{
    TickPlan plan;
    RM_TRY(prepare_tick(c, *c, plan, false, tx, n_active, first_new, src_list, src_start_us, src_air_us, air_mode, air_oldest));
    RM_TRY(launch_tick(c, *c, plan));
    if (c->ev.on && !c->draws_pending) {
        if (plan.empty && c->last_n_new > 0) {
            // a tick without receivers on this rank: its packets exist all the same (slot_off of an empty tick is not written)
            RM_HIP(hipMemsetAsync(c->d_slot_off.p, 0, (size_t(std::max(c->last.n_cnt, 0)) + 2) * sizeof(uint32_t), c->stream));
        }
        RM_TRY(ev_append(c, *c));
    }
    return RM_OK;
}

int drain_profile(rm_context *c)
{
    for (size_t i = 0; i < c->ev_used; ++i) {
        rm_context::Sample &sm = c->ev_pool[i];
        RM_HIP(hipEventSynchronize(sm.ev[sm.n]));
        for (int k = 0; k < sm.n; ++k) {
            float ms = 0;
            RM_HIP(hipEventElapsedTime(&ms, sm.ev[k], sm.ev[k + 1]));
            c->prof_ms[sm.stage[k]] += ms;
        }
        c->prof_samples++;
    }
    c->ev_used = 0;
    return RM_OK;
}

template <typename T> int upload(DevBuf<T> &d, const std::vector<T> &h, hipStream_t s)
{
    RM_HIP(d.ensure(std::max<size_t>(h.size(), 1)));
    if (!h.empty()) RM_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return RM_OK;
}

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
static const char *record_flag_message(uint32_t flag)
{
    return flag == 2u ? "a record given in device memory has a txprob strictly between 0 and 1, but no node probability asks for "
                        "java.util.Random draws: records in device memory must carry their source node's txprob"
                      : "a frame of this SINR tick lies outside the tick's [t_begin, t_end]: the batch was not self-contained";
}

bool still_on_air(const rm_tx_record &r, int64_t t_begin) { return r.start_us + r.air_us > t_begin; }

} // namespace

extern "C" {

int rm_abi_version(void) { return RM_ABI_VERSION; }

const char *rm_last_error(void) { return g_err.c_str(); }

int rm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RM_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int rm_create(int device_ordinal, rm_context **out)
{
    if (!out) return fail(RM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RM_ERR_NO_DEVICE, std::string("no HIP device available (") +
                                          (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                                          "); this engine has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(RM_ERR_INVALID, "device ordinal out of range");
    RM_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    RM_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RM_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    rm_context *c = new rm_context();
    c->device = device_ordinal;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(RM_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    c->own_stream = true;
    if (const char *g = std::getenv("RM_GRAPH")) c->use_graphs = (g[0] == '1');
    rm_model_defaults(&c->params, RM_MODEL_NULL); // Main.java:66-70: NullRadioMedium is the default
    *out = c;
    return RM_OK;
}

void rm_destroy(rm_context *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &g : c->graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto &sm : c->ev_pool)
        for (auto &e : sm.ev) (void)hipEventDestroy(e);
    c->d_x.release(); c->d_y.release(); c->d_z.release(); c->d_txpower.release(); c->d_txprob.release();
    c->d_channel.release(); c->d_int_id.release(); c->d_rx_x.release(); c->d_rx_y.release(); c->d_rx_z.release();
    c->d_rx_rxprob.release(); c->d_rx_channel.release(); c->d_rx_int_id.release(); c->d_rx_orig.release();
    c->d_pos_of.release(); c->d_rx_enabled.release(); c->d_rx_rec.release(); c->d_rx_rec32.release(); c->d_rxf.release(); c->d_bbox_xy.release();
    c->d_bbox_z.release(); c->d_wg_box_xy.release(); c->d_wg_box_z.release();
    c->d_n2n.release(); c->d_shadow_tbl.release(); c->d_air.release(); c->d_rng.release(); c->d_ticks.release();
    c->air.pool.release(); c->air.head.release(); c->air.tail.release(); c->air.mark.release(); c->air.bad.release();
    c->d_patch.release();
    c->d_enabled.release();
    (void)rm_events_disable(c);
    c->release_all();
    for (auto &sl : c->extra_slots) sl->release_all();
    for (int g = 0; g < 2; ++g) {
        if (c->h_ticks_ev[g]) (void)hipEventDestroy(c->h_ticks_ev[g]);
        if (c->h_ticks[g]) (void)hipHostFree(c->h_ticks[g]);
    }
    if (c->h_transmit) (void)hipHostFree(c->h_transmit);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_pack) (void)hipHostFree(c->h_pack);
    c->d_pack.release();
    c->d_pack_done.release();
    for (int g = 0; g < 2; ++g) {
        if (c->h_tx_ev[g]) (void)hipEventDestroy(c->h_tx_ev[g]);
        if (c->h_tx[g]) (void)hipHostFree(c->h_tx[g]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *rm_get_name(const rm_context *c) { return c ? model_name(c->params.kind) : ""; }

int rm_set_stream(rm_context *c, void *hip_stream)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    RM_HIP(hipStreamSynchronize(c->stream));
    if (c->own_stream) RM_HIP(hipStreamDestroy(c->stream));
    c->stream = static_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
    return RM_OK;
}

void rm_model_defaults(rm_model_params *p, int32_t kind)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->kind = kind;
    p->udgm_success_ratio_tx = 1.0;     // UDGMRadioMedium.java:18
    p->udgm_success_ratio_rx = 1.0;     // :20
    p->udgm_transmission_range = 50.0;  // :22
    p->udgm_interference_range = 100.0; // :24
    p->const_range = 100.0;             // UDGMConstantLossRadioMedium.java:8
    p->ld_pl0_db = 40.0;
    p->ld_exponent = 3.0;
    p->ld_d0 = 1.0;
    p->ld_sigma_db = 0.0;
    p->ld_clip = 3.0;
    p->ld_seed = 0;
    p->ld_sensitivity_dbm = -95.0;
    p->ld_noise_dbm = -100.0;           // AbstractRadioMedium.java:38
    p->ld_capture_db = 3.0;
    p->ld_ifloor_dbm = -110.0;
}

int rm_set_model(rm_context *c, const rm_model_params *p)
{
    if (!c || !p) return fail(RM_ERR_INVALID, "NULL argument");
    RM_TRY(validate_model(p));
    const bool was_geo = is_geometric(c);
    c->params = *p;
    if (was_geo != is_geometric(c)) c->rx_dirty = true;
    c->prefilter_dirty = true;
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(build_shadow_table(c));
    c->air_batches.clear();
    c->air_head = c->air_tail = 0;
    c->onair.clear();
    c->onair_tick.clear();
    c->air.valid = false;
    c->pending.clear();
    return RM_OK;
}

int rm_get_model(const rm_context *c, rm_model_params *out)
{
    if (!c || !out) return fail(RM_ERR_INVALID, "NULL argument");
    *out = c->params;
    return RM_OK;
}

int rm_set_n2n_matrix(rm_context *c, int32_t m, const double *row_major)
{
    if (!c || m < 0 || (m > 0 && !row_major)) return fail(RM_ERR_INVALID, "bad matrix");
    RM_HIP(hipSetDevice(c->device));
    c->n2n_m = m;
    if (m == 0) {
        c->d_n2n.release();
        return RM_OK;
    }
    RM_HIP(c->d_n2n.ensure(size_t(m) * m));
    RM_HIP(hipMemcpyAsync(c->d_n2n.p, row_major, size_t(m) * m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    return RM_OK;
}

int rm_set_base_rssi(rm_context *c, double rssi)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    c->base_rssi = rssi;
    return RM_OK;
}

double rm_get_base_rssi(const rm_context *c, int32_t node)
{
    (void)node; // AbstractRadioMedium.java:46-48: the same value for every node
    return c ? c->base_rssi : -100.0;
}

int rm_seed(rm_context *c, int64_t seed)
{
    return rm_set_rng_state(c, (uint64_t(seed) ^ 0x5DEECE66Dull) & ((1ull << 48) - 1));
}

int rm_set_rng_state(rm_context *c, uint64_t state48)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    RM_HIP(hipSetDevice(c->device));
    RM_HIP(c->d_rng.ensure(1));
    state48 &= (1ull << 48) - 1;
    RM_HIP(hipMemcpyAsync(c->d_rng.p, &state48, 8, hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    return RM_OK;
}

int rm_get_rng_state(rm_context *c, uint64_t *state48)
{
    if (!c || !state48) return fail(RM_ERR_INVALID, "NULL argument");
    RM_HIP(hipSetDevice(c->device));
    if (!c->d_rng.p) RM_TRY(rm_seed(c, 0));
    RM_HIP(hipMemcpyAsync(state48, c->d_rng.p, 8, hipMemcpyDeviceToHost, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    return RM_OK;
}

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
    RM_TRY(upload(c->d_channel, c->channel, c->stream));
    RM_TRY(upload(c->d_int_id, c->int_id, c->stream));
    RM_TRY(upload(c->d_enabled, c->enabled, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    if (c->ev.on) RM_TRY(ev_ensure_nodes(c));
    recompute_frame(c);
    c->rx_dirty = true;
    c->frac_probs = -1;
    c->air_batches.clear();
    c->air_head = c->air_tail = 0;
    c->onair.clear();
    c->onair_tick.clear();
    c->air.valid = false;
    c->pending.clear();
    if (c->rx_count >= 0 && c->rx_first + c->rx_count > n) {
        c->rx_first = 0;
        c->rx_count = -1;
    }
    c->rx_dirty = true;
    return RM_OK;
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
    note_probabilities(c, c->rxprob[i], c->txprob[i], rxprob, txprob);
    c->x[i] = x; c->y[i] = y; c->z[i] = z; c->txpower[i] = txpower; c->channel[i] = channel;
    c->enabled[i] = enabled; c->rxprob[i] = rxprob; c->txprob[i] = txprob;
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
    for (int k = 0; k < count; ++k) {
        const int i = nodes[k];
        c->x[i] = x[k];
        c->y[i] = y[k];
        c->z[i] = z ? z[k] : 0.0; // Position.java:44-46: set(x, y) puts z at 0
    }
    return patch_nodes(c, nodes, count);
}

int64_t rm_receiver_table_builds(const rm_context *c) { return c ? c->table_sorts : 0; }

int rm_node_count(const rm_context *c) { return c ? c->n : fail(RM_ERR_INVALID, "ctx is NULL"); }

int rm_set_partition(rm_context *c, int32_t first, int32_t count)
{
    if (!c || first < 0 || count < 0 || first + count > c->n) return fail(RM_ERR_INVALID, "partition out of range");
    c->rx_first = first;
    c->rx_count = count;
    c->rx_dirty = true;
    return RM_OK;
}

int rm_set_link_capacity(rm_context *c, uint32_t max_links)
{
    if (!c || max_links == 0) return fail(RM_ERR_INVALID, "bad capacity");
    c->cap = max_links;
    return RM_OK;
}

int rm_air_list_stats(const rm_context *c, uint64_t *incremental_ticks, uint64_t *rebuilt_ticks)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    if (incremental_ticks) *incremental_ticks = c->air.incremental;
    if (rebuilt_ticks) *rebuilt_ticks = c->air.rebuilds;
    return RM_OK;
}

int rm_set_time(rm_context *c, int64_t t)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    c->current_time = t;
    return RM_OK;
}

int64_t rm_air_time_us(int64_t hex_length) { return hex_length * 32; } // RadioPacket.java:72

void rm_event_times(int64_t start_us, int64_t air_us, int64_t current_time_us, int64_t *t_start, int64_t *t_end)
{
    int64_t packetTime = start_us; // Simulator.java:323-326
    if (packetTime < current_time_us) packetTime = current_time_us;
    if (t_start) *t_start = packetTime;
    if (t_end) *t_end = packetTime + air_us;
}

int rm_tick_begin(rm_context *c, int64_t t_begin_us, int64_t t_end_us)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    c->t_begin = t_begin_us;
    c->t_end = t_end_us;
    c->pending.clear();
    c->tick_frac_records = false;
    if (is_sinr(c)) {
        size_t k = 0;
        c->onair_tick.resize(c->onair.size(), 0u);
        for (size_t i = 0; i < c->onair.size(); ++i)
            if (still_on_air(c->onair[i], t_begin_us)) {
                c->onair_tick[k] = c->onair_tick[i];
                c->onair[k++] = c->onair[i];
            }
        c->onair.resize(k);
        c->onair_tick.resize(k);
    } else {
        c->onair.clear();
        c->onair_tick.clear();
        c->air.valid = false;
    }
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

static int copy_out(rm_context *c, TickSlot &ts, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr,
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

static size_t pad64(size_t v) { return (v + 63) & ~size_t(63); }

static rm::HostView stage_view(char *base, uint32_t links, uint32_t packets, size_t *bytes)
{
    rm::HostView v{};
    size_t o = 0;
    v.hdr = reinterpret_cast<rm::HostHeader *>(base + o); o += pad64(sizeof(rm::HostHeader));
    o += pad64(sizeof(rm::BatchCounts) * RM_MAX_BATCH); // per-slot counts of rm_batch_result_view (stage_counts)
    v.pkt_offset = reinterpret_cast<uint32_t *>(base + o); o += pad64((size_t(packets) + 1) * 4);
    v.pkt_interference = reinterpret_cast<uint8_t *>(base + o); o += pad64(size_t(packets) + 1);
    v.pkt = reinterpret_cast<int32_t *>(base + o); o += pad64(size_t(links) * 4);
    v.dst = reinterpret_cast<int32_t *>(base + o); o += pad64(size_t(links) * 4);
    v.rssi = reinterpret_cast<double *>(base + o); o += pad64(size_t(links) * 8);
    v.sinr = reinterpret_cast<double *>(base + o); o += pad64(size_t(links) * 8);
    v.verdict = reinterpret_cast<uint8_t *>(base + o); o += pad64(size_t(links) + 1);
    v.links = links;
    v.packets = packets;
    if (bytes) *bytes = o;
    return v;
}

static rm::BatchCounts *stage_counts(char *base) { return reinterpret_cast<rm::BatchCounts *>(base + pad64(sizeof(rm::HostHeader))); }

static int ensure_stage(rm_context *c, uint32_t links, uint32_t packets)
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

// pack the evaluated tick of slot `ts` into the host-mapped block and wait for it
static int pack_to_stage(rm_context *c, TickSlot &ts, rm::HostView *view)
{
    if (ts.draws_pending)
        return fail(RM_ERR_STATE, "this rank's verdicts wait for the other ranks' draw counts: exchange "
                                  "rm_draw_counts_device and call rm_tick_finish_draws first");
    const int n_new = std::max(ts.last_n_new, 0);
    const int have_offsets = (n_new > 0 && part_count(c) > 0) ? 1 : 0;
    RM_TRY(ensure_stage(c, 0, uint32_t(n_new)));
    for (int attempt = 0; attempt < 2; ++attempt) {
        const rm::HostView v = stage_view(c->h_stage, c->stage_links, c->stage_packets, nullptr);
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

static int tick_run_host(rm_context *c);

int rm_tick_run(rm_context *c)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    return tick_run_host(c);
}

static int stage_status(rm_context *c, const rm::HostView &v)
{
    if (v.hdr->span_flag) return fail(RM_ERR_STATE, record_flag_message(v.hdr->span_flag));
    if (v.hdr->dropped) {
        c->air.valid = false;
        return fail(RM_ERR_CAPACITY, "heard links exceed the context's link capacity (rm_set_link_capacity)");
    }
    return RM_OK;
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
    out->pkt = v.pkt;
    out->dst = v.dst;
    out->verdict = v.verdict;
    out->rssi = v.rssi;
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
        if (pkt) std::memcpy(pkt, v.pkt, k * sizeof(int32_t));
        if (dst) std::memcpy(dst, v.dst, k * sizeof(int32_t));
        if (verdict) std::memcpy(verdict, v.verdict, k);
        if (rssi) std::memcpy(rssi, v.rssi, k * sizeof(double));
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

// evaluate the tick enqueued with rm_tick_begin / rm_enqueue_tx* (results stay on the device)
static int tick_run_host(rm_context *c)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    if (!c->in_tick) return fail(RM_ERR_STATE, "rm_tick_flush without rm_tick_begin");
    RM_HIP(hipSetDevice(c->device));
    c->in_tick = false;
    // SINR: the frames of earlier ticks have their entries in the lists on the device -- only the new frames go there
    // (and are evaluated), unless the lists have to be rebuilt from everything on the air
    const bool sinr = is_sinr(c);
    int air_mode = kAirNone;
    uint32_t oldest = 0;
    if (sinr) {
        for (const rm_tx_record &r : c->pending)
            if (r.air_us < 0 || r.air_us > int64_t(UINT32_MAX))
                return fail(RM_ERR_INVALID, "a frame of the SINR medium has to be shorter than 2^32 us");
        c->onair_tick.resize(c->onair.size(), 0u);
        for (uint32_t k : c->onair_tick) oldest = (oldest == 0 || k < oldest) ? k : oldest;
        const bool unknown = std::find(c->onair_tick.begin(), c->onair_tick.end(), 0u) != c->onair_tick.end();
        air_mode = (!unknown && air_lists_current(c, c->t_begin, oldest)) ? kAirIncremental : kAirRebuild;
    }
    const size_t n_old = (air_mode == kAirIncremental) ? 0 : c->onair.size();
    const int first_new = int(n_old);
    const size_t total = n_old + c->pending.size();
    RM_HIP(c->d_tx.ensure(std::max<size_t>(total, 1)));
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
            RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_tx[g]), want * sizeof(rm_tx_record), hipHostMallocDefault));
            c->h_tx_n[g] = want;
        }
        if (n_old) std::memcpy(c->h_tx[g], c->onair.data(), n_old * sizeof(rm_tx_record));
        if (!c->pending.empty()) std::memcpy(c->h_tx[g] + n_old, c->pending.data(), c->pending.size() * sizeof(rm_tx_record));
        RM_HIP(hipMemcpyAsync(c->d_tx.p, c->h_tx[g], total * sizeof(rm_tx_record), hipMemcpyHostToDevice, c->stream));
        RM_HIP(hipEventRecord(c->h_tx_ev[g], c->stream));
    }
    const int rc = run_tick(c, c->d_tx.p, int(total), first_new, nullptr, 0, 0, air_mode, oldest);
    c->tick_frac_records = false;
    if (rc != RM_OK) {
        c->air.valid = false;
        return rc;
    }
    if (sinr) {
        if (air_mode == kAirRebuild) std::fill(c->onair_tick.begin(), c->onair_tick.end(), c->air.tick);
        c->onair.insert(c->onair.end(), c->pending.begin(), c->pending.end());
        c->onair_tick.resize(c->onair.size(), c->air.tick);
    }
    c->pending.clear();
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
    c->onair.clear();
    c->onair_tick.clear();
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

static int air_tick_device(rm_context *c, int64_t t_begin_us, const int32_t *dev_src, const rm_tx_record *dev_new, int32_t n,
                           int64_t start_us, int64_t air_us, int64_t latest_end_us);

int rm_tick_run_sources_device(rm_context *c, int64_t t_begin_us, int64_t t_end_us, const int32_t *dev_src, int32_t n,
                               int64_t start_us, int64_t air_us)
{
    if (!c || n < 0 || (n > 0 && !dev_src) || air_us < 0) return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    c->t_begin = t_begin_us;
    c->t_end = t_end_us;
    if (!is_sinr(c)) {
        RM_HIP(c->d_tx.ensure(std::max(n, 1)));
        return run_tick(c, c->d_tx.p, n, 0, dev_src, start_us, air_us);
    }
    return air_tick_device(c, t_begin_us, dev_src, nullptr, n, start_us, air_us, start_us + air_us);
}

// The SINR medium's tick with its frames in device memory -- built from source indices (dev_src: all with the same start
// and air time) or given as records (dev_new; `latest_end_us` bounds their start + air: the host never reads them).
// The frames of earlier calls that are still on the air stay resident on the device (the window [air_head, air_tail) of
// d_air): a tick that only adds frames sweeps the new ones, a rebuild of the on-air lists sweeps the whole window.
static int air_tick_device(rm_context *c, int64_t t_begin_us, const int32_t *dev_src, const rm_tx_record *dev_new, int32_t n,
                           int64_t start_us, int64_t air_us, int64_t latest_end_us)
{
    // Expire whole batches (rm_tick_begin's rule: start + air > t_begin stays).
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
    // room for the new batch at the tail; slide the window to the front when the buffer is used up
    const size_t live = c->air_tail - c->air_head;
    if (c->air_tail + size_t(n) > c->d_air.n) {
        const size_t want = std::max<size_t>(4 * (live + size_t(n)), 1 << 16);
        DevBuf<rm_tx_record> fresh;
        RM_HIP(fresh.ensure(want));
        if (live)
            RM_HIP(hipMemcpyAsync(fresh.p, c->d_air.p + c->air_head, live * sizeof(rm_tx_record), hipMemcpyDeviceToDevice, c->stream));
        RM_HIP(hipStreamSynchronize(c->stream));
        c->d_air.release();
        c->d_air = fresh;
        c->air_head = 0;
        c->air_tail = live;
    }
    if (dev_src && air_us > int64_t(UINT32_MAX)) return fail(RM_ERR_INVALID, "a frame of the SINR medium has to be shorter than 2^32 us");
    if (dev_new && n > 0) // the caller's records join the window (they have to be there when the lists are rebuilt)
        RM_HIP(hipMemcpyAsync(c->d_air.p + c->air_tail, dev_new, size_t(n) * sizeof(rm_tx_record), hipMemcpyDeviceToDevice, c->stream));
    uint32_t oldest = 0;
    bool unknown = false;
    for (const auto &bt : c->air_batches) {
        unknown = unknown || bt.tick == 0;
        oldest = (oldest == 0 || bt.tick < oldest) ? bt.tick : oldest;
    }
    const int air_mode = (!unknown && air_lists_current(c, t_begin_us, oldest)) ? kAirIncremental : kAirRebuild;
    // the records of the new frames are built at the window's tail either way; an incremental tick sweeps only those
    const int first_new = (air_mode == kAirIncremental) ? 0 : int(live);
    const rm_tx_record *base = c->d_air.p + c->air_head + (air_mode == kAirIncremental ? live : 0);
    const int rc = run_tick(c, base, first_new + n, first_new, dev_src, start_us, air_us, air_mode, oldest);
    if (rc != RM_OK) {
        c->air.valid = false;
        return rc;
    }
    if (air_mode == kAirRebuild)
        for (auto &bt : c->air_batches) bt.tick = c->air.tick;
    c->air_tail += size_t(n);
    if (n > 0) c->air_batches.push_back({n, latest_end_us, c->air.tick});
    return RM_OK;
}

int rm_tick_run_records_device(rm_context *c, int64_t t_begin_us, int64_t t_end_us, const rm_tx_record *dev_new, int32_t n_new,
                               int64_t latest_end_us)
{
    if (!c || n_new < 0 || (n_new > 0 && !dev_new)) return fail(RM_ERR_INVALID, "bad arguments");
    if (!is_sinr(c)) return rm_tick_run_device(c, t_begin_us, t_end_us, dev_new, n_new);
    RM_HIP(hipSetDevice(c->device));
    c->t_begin = t_begin_us;
    c->t_end = t_end_us;
    return air_tick_device(c, t_begin_us, nullptr, dev_new, n_new, 0, 0, latest_end_us);
}

static int result_device(rm_context *c, TickSlot &ts, rm_device_result *out)
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

int rm_tick_finish_draws(rm_context *c, const uint32_t *all_counts, int32_t world, int32_t rank, int on_device)
{
    if (!c || !all_counts || world < 1 || rank < 0 || rank >= world) return fail(RM_ERR_INVALID, "bad arguments");
    if (!c->draws_pending) return fail(RM_ERR_STATE, "no tick is waiting for draw counts");
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

static int result_count(rm_context *c, TickSlot &ts, uint32_t *count, uint32_t *dropped)
{
    if (!ts.have_result) return fail(RM_ERR_STATE, "no evaluated tick");
    RM_HIP(hipSetDevice(c->device));
    RM_TRY(materialize(c, ts));
    uint32_t oc[5];
    RM_HIP(hipMemcpyAsync(oc, ts.last.out_count, sizeof(oc), hipMemcpyDeviceToHost, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    if (oc[4]) return fail(RM_ERR_STATE, record_flag_message(oc[4]));
    if (count) *count = oc[2];
    if (dropped) *dropped = oc[1];
    return RM_OK;
}

int rm_result_count(rm_context *c, uint32_t *count, uint32_t *dropped)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    return result_count(c, *c, count, dropped);
}

// ---- several independent ticks per pass ------------------------------------------------------------

static TickSlot *slot_of(rm_context *c, int32_t slot)
{
    if (slot == 0) return c;
    if (slot < 0 || size_t(slot) > c->extra_slots.size()) return nullptr;
    return c->extra_slots[size_t(slot) - 1].get();
}

// the launch sequence of n prepared ticks in four launches (sorted table, fp32 frame, no SINR)
static int launch_batch(rm_context *c, TickSlot *const *slots, const TickPlan *plans, int n)
{
    rm::TickDev ticks[RM_MAX_BATCH];
    for (int b = 0; b < n; ++b) ticks[b] = plans[b].t;
    // the descriptors go to device memory (k_store_ticks, ordered on the stream after the previous
    // batch's kernels, which read the same array)
    RM_HIP(c->d_ticks.ensure(RM_MAX_BATCH));
    rm::TickDev *dev_ticks = c->d_ticks.p;
    const bool by_copy = n > 2 * 6; // beyond two k_store_ticks launches: one fetch from pinned, host-mapped memory
    if (by_copy) {
        const int g = c->h_ticks_gen;
        c->h_ticks_gen ^= 1;
        if (!c->h_ticks[g]) {
            RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_ticks[g]), sizeof(rm::TickDev) * RM_MAX_BATCH, hipHostMallocMapped));
            RM_HIP(hipEventCreateWithFlags(&c->h_ticks_ev[g], hipEventDisableTiming));
        } else {
            RM_HIP(hipEventSynchronize(c->h_ticks_ev[g]));
        }
        std::memcpy(c->h_ticks[g], ticks, sizeof(rm::TickDev) * size_t(n));
        RM_HIP(rm::launch_fetch_ticks(c->stream, c->h_ticks[g], n, dev_ticks)); // the device reads the mapped block itself
        RM_HIP(hipEventRecord(c->h_ticks_ev[g], c->stream));
    }
    const rm::ModelDev m = model_dev(c);
    const rm::NodesDev nd = nodes_dev(c);
    const rm::LaunchCfg &cfg = plans[0].cfg;
    hipStream_t s = c->stream;
    const bool sample = c->profile && (c->tick_index++ % uint64_t(c->profile_every) == 0);
    rm_context::Sample *smp = nullptr;
    if (sample) {
        if (c->ev_used == c->ev_pool.size()) {
            rm_context::Sample ns;
            for (auto &e : ns.ev) RM_HIP(hipEventCreate(&e));
            c->ev_pool.push_back(ns);
        }
        smp = &c->ev_pool[c->ev_used++];
        smp->n = 0;
    }
    auto stage = [&](int id) -> int {
        if (smp) {
            RM_HIP(hipEventRecord(smp->ev[smp->n], s));
            smp->stage[smp->n++] = id;
        }
        return RM_OK;
    };
    if (!by_copy) RM_HIP(rm::launch_store_ticks(s, ticks, n, dev_ticks));
    if (smp) RM_TRY(stage(RM_STAGE_EMPTY)); // calibration: an empty bracket
    // RM_BATCH_FRAMES=1: the batch through the one-frame-per-workgroup kernel of the closed-loop tick instead of the
    // three sweep stages (one launch; the compact arrays on demand, per slot)
    static const bool batch_frames = std::getenv("RM_BATCH_FRAMES") != nullptr;
    if (batch_frames && !cfg.stochastic && !plans[0].sinr) {
        int seg_len = rm::frame_tick_segment(ticks[0], cfg, m);
        for (int b = 1; b < n && seg_len > 0; ++b) seg_len = std::min(seg_len, rm::frame_tick_segment(ticks[b], cfg, m));
        if (seg_len > 0) {
            RM_TRY(stage(RM_STAGE_FILTER));
            RM_HIP(rm::launch_tick_frames_batch(s, nd, m, ticks, n, dev_ticks, cfg, seg_len));
            if (smp) RM_HIP(hipEventRecord(smp->ev[smp->n], s));
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
    RM_TRY(stage(RM_STAGE_EXACT));
    RM_HIP(rm::launch_batch_stage(s, 1, nd, m, ticks, n, dev_ticks, cfg));
    if (plans[0].sinr) {
        RM_TRY(stage(RM_STAGE_SINR));
        RM_HIP(rm::launch_batch_stage(s, 3, nd, m, ticks, n, dev_ticks, cfg));
    }
    RM_TRY(stage(RM_STAGE_REORDER));
    RM_HIP(rm::launch_batch_stage(s, 2, nd, m, ticks, n, dev_ticks, cfg));
    if (cfg.stochastic) {
        // the shared generator is walked tick by tick, in slot order, inside one launch
        RM_TRY(stage(RM_STAGE_DRAWS));
        RM_HIP(rm::launch_draws_batch(s, m, ticks, n, dev_ticks));
    }
    if (smp) RM_HIP(hipEventRecord(smp->ev[smp->n], s));
    for (int b = 0; b < n; ++b) {
        slots[b]->have_result = true;
        slots[b]->compact_pending = false;
    }
    return RM_OK;
}

static int batch_run(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us,
                     const int32_t *const *dev_src, const rm_tx_record *const *dev_new, const int32_t *n_per,
                     const int64_t *start_us, const int64_t *air_us)
{
    if (!c || n_ticks < 1 || n_ticks > RM_MAX_BATCH || !t_begin_us || !t_end_us || !n_per || (!dev_src && !dev_new) ||
        (dev_src && (!start_us || !air_us)))
        return fail(RM_ERR_INVALID, "bad arguments");
    for (int b = 0; b < n_ticks; ++b)
        if (n_per[b] < 0 || (n_per[b] > 0 && !(dev_src ? (const void *)dev_src[b] : (const void *)dev_new[b])) ||
            (dev_src && air_us[b] < 0))
            return fail(RM_ERR_INVALID, "bad arguments");
    RM_HIP(hipSetDevice(c->device));
    const bool sinr = is_sinr(c);
    if (sinr) {
        // The SINR extension looks at every frame on the air.  A batch is accepted when its ticks are
        // self-contained: nothing of an earlier call and nothing of an earlier tick of the batch is
        // still on the air when a tick begins (e.g. air time <= tick length).  Frames given as source
        // indices carry their time span in the arguments; records given by the caller (the gathered
        // records of a multi-GPU batch) are verified on the device: every frame of tick b has to lie
        // inside [t_begin[b], t_end[b]], and the ticks must not overlap.
        if (!dev_src)
            for (int b = 0; b < n_ticks; ++b)
                if (t_end_us[b] < t_begin_us[b] || (b + 1 < n_ticks && t_end_us[b] > t_begin_us[b + 1]))
                    return fail(RM_ERR_STATE, "SINR batches of records need ticks [t_begin, t_end] that do not overlap");
        for (const auto &bt : c->air_batches)
            if (bt.end_us > t_begin_us[0])
                return fail(RM_ERR_STATE, "frames of earlier calls are still on the air: run this tick on its own");
        for (const auto &r : c->onair)
            if (still_on_air(r, t_begin_us[0]))
                return fail(RM_ERR_STATE, "frames of earlier calls are still on the air: run this tick on its own");
        for (int b = 0; dev_src && b + 1 < n_ticks; ++b)
            if (n_per[b] > 0 && start_us[b] + air_us[b] > t_begin_us[b + 1])
                return fail(RM_ERR_STATE, "the SINR medium carries frames that outlive their tick into the next one: run "
                                          "overlapping ticks one at a time");
        c->onair.clear();
        c->onair_tick.clear();
        c->air.valid = false; // the ticks of a batch keep their lists to themselves
        c->air_batches.clear();
        c->air_head = c->air_tail = 0;
    }
    if (maybe_draws(c) && part_count(c) != c->n)
        return fail(RM_ERR_STATE, "a receiver partition whose links draw needs rm_tick_finish_draws per tick: run it one "
                                  "tick at a time");
    while (c->extra_slots.size() + 1 < size_t(n_ticks)) c->extra_slots.emplace_back(new TickSlot());
    const rm::PlanKnobs knobs = rm::read_plan_knobs(); // once for the whole batch
    TickSlot *slots[RM_MAX_BATCH];
    TickPlan plans[RM_MAX_BATCH];
    bool batched = true;
    for (int b = 0; b < n_ticks; ++b) {
        TickSlot &ts = *slot_of(c, b);
        slots[b] = &ts;
        const rm_tx_record *tx = nullptr;
        if (dev_src && sinr && b == n_ticks - 1) {
            // the last tick's frames may outlive the batch: they are built where the on-air list of the
            // one-tick-at-a-time path lives
            RM_HIP(c->d_air.ensure(std::max<size_t>(size_t(n_per[b]), 1 << 16)));
            tx = c->d_air.p;
        } else if (dev_src) {
            RM_HIP(ts.d_tx.ensure(std::max(n_per[b], 1)));
            tx = ts.d_tx.p;
        } else {
            tx = dev_new[b];
        }
        c->dev_records_from_caller = (dev_src == nullptr);
        const int rc_prep = prepare_tick(c, ts, plans[b], true, tx, n_per[b], 0, dev_src ? dev_src[b] : nullptr,
                                         dev_src ? start_us[b] : 0, dev_src ? air_us[b] : 0, kAirNone, 0, &knobs);
        c->dev_records_from_caller = false;
        RM_TRY(rc_prep);
        batched = batched && !plans[b].empty && rm::batch_eligible(plans[b].t, plans[b].cfg, model_dev(c)) &&
                  plans[b].t.rpt == plans[0].t.rpt;
    }
    c->t_begin = t_begin_us[0];
    c->t_end = t_end_us[n_ticks - 1];
    if (sinr && !dev_src)
        for (int b = 0; b < n_ticks; ++b) {
            plans[b].t.check_span = 1;
            plans[b].t.span_begin = t_begin_us[b];
            plans[b].t.span_end = t_end_us[b];
        }
    if (sinr && dev_src && n_per[n_ticks - 1] > 0) {
        c->air_tail = size_t(n_per[n_ticks - 1]);
        c->air_batches.push_back({n_per[n_ticks - 1], start_us[n_ticks - 1] + air_us[n_ticks - 1], 0u});
    }
    if (batched) {
        if (sinr)
            for (int b = 0; b < n_ticks; ++b) plans[b].t.reset_heads = 1;
        return launch_batch(c, slots, plans, n_ticks);
    }
    // configurations the batched kernels do not cover (fp64 frame, unsorted table, very many frames,
    // empty ticks): the same ticks, one launch sequence each
    for (int b = 0; b < n_ticks; ++b) RM_TRY(launch_tick(c, *slots[b], plans[b]));
    return RM_OK;
}

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
    for (int attempt = 0;; ++attempt) {
        v = stage_view(c->h_stage, c->stage_links, c->stage_packets, nullptr);
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
        r.pkt = v.pkt + bc.link_base;
        r.dst = v.dst + bc.link_base;
        r.verdict = v.verdict + bc.link_base;
        r.rssi = v.rssi + bc.link_base;
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

// ---- several devices behind one caller --------------------------------------------------------------------

struct rm_group {
    std::vector<rm_context *> m;
    int n_nodes = 0;
    int n_new = 0; // frames of the running tick
    bool in_tick = false;
    std::vector<uint32_t> counts; // [world][n_new] per-packet draw counts
};

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
    for (rm_context *c : g->m) rm_destroy(c);
    delete g;
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
        RM_TRY(rm_set_partition(c, lo, hi - lo)); // receivers range-partitioned by node index
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

int rm_group_tick_flush(rm_group *g, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr, uint32_t cap,
                        uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset)
{
    if (!g || !g->in_tick) return fail(RM_ERR_STATE, "rm_group_tick_flush without rm_group_tick_begin");
    g->in_tick = false;
    const int world = int(g->m.size());
    const int n_new = g->n_new;
    // every member's launches are enqueued before anything is waited for
    for (rm_context *c : g->m) RM_TRY(tick_run_host(c));
    // probabilistic links: the members' per-packet draw counts, through the host
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
        for (int r = 0; r < world; ++r) RM_TRY(rm_tick_finish_draws(g->m[size_t(r)], g->counts.data(), world, r, 0));
    }
    // the members' results in their pinned blocks, then merged packet by packet in member (= node) order
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
        o.pkt = v.pkt;
        o.dst = v.dst;
        o.verdict = v.verdict;
        o.rssi = v.rssi;
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
    for (int q = 0; q < n_new; ++q) {
        if (pkt_offset) pkt_offset[q] = w;
        for (int r = 0; r < world; ++r) {
            const rm_host_result &o = res[size_t(r)];
            if (uint32_t(q) >= o.n_packets) continue;
            const uint32_t b = o.pkt_offset[q], e = std::min(o.pkt_offset[q + 1], o.count);
            for (uint32_t i = b; i < e; ++i, ++w) {
                if (w >= cap) continue;
                if (pkt) pkt[w] = q;
                if (dst) dst[w] = o.dst[i];
                if (verdict) verdict[w] = o.verdict[i];
                if (rssi) rssi[w] = o.rssi[i];
                if (sinr) sinr[w] = o.sinr ? o.sinr[i] : 0.0;
            }
        }
        // the packet-level Tx-failure flag is the same on every member (one generator, one draw)
        if (pkt_interference && world > 0 && uint32_t(q) < res[0].n_packets) pkt_interference[q] = res[0].pkt_interference[q];
    }
    if (pkt_offset) pkt_offset[std::max(n_new, 0)] = w;
    if (first_error != RM_OK) return fail(first_error, first_msg);
    if (total > cap) return fail(RM_ERR_CAPACITY, "caller buffers too small for the heard links");
    return RM_OK;
}

// ---- reception stage ------------------------------------------------------------------------------------

static size_t ev_out_bytes(uint32_t cap)
{
    return pad64(sizeof(rm::EvHeader)) + pad64(size_t(cap) * 8) + pad64(size_t(cap) * 4) + pad64(size_t(cap) * 8);
}

static rm::EvOut ev_out(rm_context *c)
{
    rm::EvOut o{};
    char *b = c->ev.h_out;
    const uint32_t cap = c->ev.pool_cap;
    size_t off = 0;
    o.hdr = reinterpret_cast<rm::EvHeader *>(b + off); off += pad64(sizeof(rm::EvHeader));
    o.pkt = reinterpret_cast<int64_t *>(b + off); off += pad64(size_t(cap) * 8);
    o.dst = reinterpret_cast<int32_t *>(b + off); off += pad64(size_t(cap) * 4);
    o.rssi = reinterpret_cast<double *>(b + off);
    o.cap = cap;
    return o;
}

static uint32_t pow2_at_least(uint32_t v)
{
    uint32_t p = 64;
    while (p < v && p < (1u << 30)) p <<= 1;
    return p;
}

int rm_events_disable(rm_context *c)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    rm_context::Events &v = c->ev;
    if (v.on || v.h_out) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    v.d_st.release(); v.d_pk.release(); v.d_ldst.release(); v.d_lrssi.release(); v.d_lverdict.release();
    v.d_gtime.release(); v.d_gmeta.release(); v.d_gref.release(); v.d_grank.release(); v.d_cnt.release(); v.d_off.release();
    v.d_recv_key.release(); v.d_send_key.release(); v.d_receiving.release(); v.d_sending.release(); v.d_latched.release();
    v.d_info_nodes.release();
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
    rm::EvState st{};
    st.top_max = int64_t(0x8000000000000000ull); // the top list is empty
    st.first_live = 0xFFFFFFFFu;
    RM_HIP(hipMemcpyAsync(v.d_st.p, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
    RM_HIP(hipStreamSynchronize(c->stream));
    RM_HIP(hipHostMalloc(reinterpret_cast<void **>(&v.h_out), ev_out_bytes(v.pool_cap), hipHostMallocMapped));

    std::memset(v.h_out, 0, pad64(sizeof(rm::EvHeader)));
    v.seq = 0;
    v.on = true;
    v.next_packet = 0;
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
    RM_HIP(rm::launch_ev_drain(c->stream, ev_dev(c), o, time_us, seq));
    c->current_time = time_us; // Simulator.java:156
    volatile const uint32_t *flag = &o.hdr->seq;
    bool seen = false;
    for (int spin = 0; spin < 400000 && !seen; ++spin) seen = (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq);
    if (!seen) RM_HIP(hipStreamSynchronize(c->stream));
    out->count = o.hdr->count;
    out->pending_packets = o.hdr->pending_packets;
    out->packet = o.pkt;
    out->dst = o.dst;
    out->rssi = o.rssi;
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

double rm_det_math(int32_t fn, double x) { return rm::host_det_math(fn, x); }

uint64_t rm_link_hash(uint64_t seed, uint32_t a, uint32_t b, double *u) { return rm::host_link_hash(seed, a, b, u); }

void rm_evq_init(rm_evq_order *o)
{
    if (!o) return;
    o->top_start = 0; // EventQueue.java:51
    o->top_max = 0;
    o->ladders = 0;
    o->top_nonempty = 0;
}

int32_t rm_evq_add(rm_evq_order *o, int64_t time_us)
{
    rm::EvOrder e{o->top_start, o->top_max, o->ladders, o->top_nonempty};
    const int32_t lad = rm::ev_ladder(e, time_us);
    rm::ev_note_top(e, time_us);
    o->top_max = e.top_max;
    o->top_nonempty = e.top_nonempty;
    return lad;
}

void rm_evq_drain(rm_evq_order *o, int64_t time_us)
{
    rm::EvOrder e{o->top_start, o->top_max, o->ladders, o->top_nonempty};
    rm::ev_drain(e, time_us);
    o->top_start = e.top_start;
    o->top_max = e.top_max;
    o->ladders = e.ladders;
    o->top_nonempty = e.top_nonempty;
}

int rm_sync(rm_context *c)
{
    if (!c) return fail(RM_ERR_INVALID, "ctx is NULL");
    RM_HIP(hipSetDevice(c->device));
    RM_HIP(hipStreamSynchronize(c->stream));
    return RM_OK;
}

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

uint64_t rm_lcg_jump(uint64_t state48, uint64_t steps)
{
    uint64_t A, C;
    rm::host_lcg_jump_map(steps, &A, &C);
    return (A * state48 + C) & ((1ull << 48) - 1);
}

double rm_lcg_next_double(uint64_t *state48)
{
    const uint64_t a = 0x5DEECE66Dull, cc = 0xBull, mask = (1ull << 48) - 1;
    uint64_t s = *state48;
    s = (s * a + cc) & mask;
    const int64_t hi = int64_t(s >> 22);
    s = (s * a + cc) & mask;
    const int64_t lo = int64_t(s >> 21);
    *state48 = s;
    return double((hi << 27) + lo) * 0x1.0p-53;
}

} // extern "C"
