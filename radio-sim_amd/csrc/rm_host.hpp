// rm_host.hpp -- what the C-ABI translation units (rm_api_*.cpp) share: the context, its result slots, the
// helpers that plan and launch a tick.  Internal: not part of the boundary (include/radiomedium_hip.h).
//
// There is deliberately no CPU fallback anywhere behind this header: every evaluation goes through the gfx950
// kernels of rm_*.hip, and rm_create fails when no HIP device can be used.
//
// Reference paths: /root/reference/radio-medium/java/se/sics/emul8/radiomedium/.
#pragma once

#include "rm_engine.h"
#include "rm_evorder.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace rmh {

extern thread_local std::string g_err; // rm_last_error()
int fail(int code, const std::string &msg);

#define RM_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return rmh::fail(RM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
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

} // namespace rmh

using rmh::DevBuf;

// Everything one evaluated tick owns on the device.  A context is its own slot 0; rm_batch_*
// adds further slots so that several ticks can be in flight through one launch sequence.
struct TickSlot {
    DevBuf<rm_tx_record> d_tx;   // records uploaded by the host / built from source indices
    DevBuf<float4> d_p_txf;      // per-frame pre-filter records
    DevBuf<int32_t> d_p_ch, d_p_src;
    DevBuf<float> d_p_inv;

    DevBuf<uint32_t> d_cnt, d_off, d_slot_tot, d_slot_off;
    // a rank's frame list (k_rank_frames): the listed frames' gathered-slot numbers, the listed frames before
    // every gathered slot, and the tick's LOCAL offsets (d_slot_off then holds the offsets by global packet number)
    DevBuf<int32_t> d_fl_map;
    DevBuf<uint32_t> d_fl_lb, d_slot_off_loc;
    DevBuf<unsigned long long> d_dense_mask; // the dense tick's heard links: 16 lane masks per (frame, chunk of 1024 nodes)

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
    DevBuf<float4> d_scan_xyzr;  // the SINR medium's tick by scan: one record per frame on the air (TickDev::scan_xyzr / scan_ch) ...
    DevBuf<int32_t> d_scan_ch;
    DevBuf<uint32_t> d_sg_cnt, d_sg_every; // ... and the tick's index of them (ScanDev): counters in two parities, the cells' entries
    DevBuf<float4> d_sg_bxyzr;
    DevBuf<int2> d_sg_bci;
    DevBuf<int32_t> d_self_next;
    DevBuf<unsigned long long> d_self_slot; // per node; entries carry the tick's stamp (never cleared)
    int sg_parity = 0;
    bool sg_clean[2] = {false, false};      // the parity's counters are zero (the tick by scan before left them so)
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
    // the dense tick (rm_dense.hip) leaves the heard links as lane masks per (frame, 1024 nodes) cell: the records are written
    // when somebody asks for them (materialize); rm_result_dense hands out the masks themselves
    bool dense_pending = false, dense_result = false;
    bool dense_layout_pending = false; // ... and so are the cells' offsets and the totals (dense_layout)
    int dense_rx_first = 0, dense_chunks = 0;
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
    DevBuf<double> d_x, d_y, d_z, d_txpower, d_txprob, d_rxprob_node;
    DevBuf<int32_t> d_channel, d_int_id;
    DevBuf<rm::SrcRecord> d_srec; // ... and the same as one 64-byte record per node (what a packet copies from its source)
    // device-resident receiver table of this partition (SoA, engine order = spatially sorted)
    DevBuf<double> d_rx_x, d_rx_y, d_rx_z, d_rx_rxprob;
    DevBuf<int32_t> d_rx_channel, d_rx_int_id, d_rx_orig, d_pos_of;
    DevBuf<uint8_t> d_rx_enabled;
    DevBuf<rm::RxRecord> d_rx_rec;
    DevBuf<rm::RxCompact> d_rx_rec32;
    DevBuf<float4> d_rxf, d_bbox_xy, d_wg_box_xy;
    DevBuf<uint32_t> d_grp_chmask, d_wg_chmask;
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
    std::vector<int32_t> h_pos_of;  // node index - pos_first -> engine position, -1 = no receiver here (host copy of d_pos_of)
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
        DevBuf<uint32_t> d_gref, d_grank, d_cnt, d_off, d_grun;
        DevBuf<unsigned long long> d_run_rec;
        DevBuf<unsigned long long> d_recv_key, d_send_key;
        DevBuf<uint8_t> d_receiving, d_sending;
        DevBuf<double> d_latched;
        int state_n = 0;          // nodes the radio-state arrays hold
        char *h_out = nullptr;    // host-mapped: EvHeader + packet / dst / rssi arrays of pool_cap entries
        char *h_info = nullptr;   // host-mapped: node-info answers
        int info_n = 0;
        uint32_t seq = 0, info_seq = 0;
        DevBuf<int32_t> d_info_nodes;
        // rm_node_info_changed: what was reported last per node, the kernel's counters, the host-mapped block of the changes
        DevBuf<double> d_rep_rssi;
        DevBuf<int2> d_rep_sc;
        DevBuf<uint32_t> d_rep_cnt;
        int rep_n = 0;            // nodes the report arrays hold (all "never reported" when (re)allocated)
        char *h_changed = nullptr;
        int changed_n = 0;
        uint32_t changed_seq = 0;
        int64_t next_packet = 0;  // host mirror of EvTails::gseq_next
        int64_t oldest_packet = 0; // number of the oldest pending packet as the last drain published it (a bound on the ring window)
        int par = 0;              // which EvState::tails are current (flips with every appended tick)
        // the tick evaluated last, not handed to the rings yet: a drain that follows at once takes it along (append + selection in
        // one launch, rm_events.hip); anything else that comes first -- another tick above all -- appends it on its own
        struct Pending {
            bool on = false;
            rm::EvLinkSrc ls{};
            const rm_tx_record *tx = nullptr;
            int n_new = 0, immediate = 0;
            int64_t now = 0;
            const uint32_t *dropped = nullptr;
        } pending;
    } ev;
    DevBuf<uint8_t> d_enabled;   // Transciever.isEnabled by node index

    int frac_probs = -1;         // cached: any rx/tx probability strictly between 0 and 1 (-1 = unknown)
    bool tick_frac_records = false; // a host record of the running tick has 0 < txprob < 1 (rm_enqueue_tx_records)
    bool rx_dirty = true;        // receiver table has to be rebuilt (positions / partition / model class)
    bool prefilter_dirty = true; // pre-filter records have to be recomputed
    double org[3] = {0, 0, 0};
    double coord_bound = 0, f32_slack = 0;

    int rx_first = 0, rx_count = -1; // receiver partition by index range (rm_set_partition); -1 = all nodes
    // receiver partition by region (rm_set_partition_spatial): region sp_part of sp_parts of the k-d split over ALL nodes;
    // sp_parts == 0: none.  The member nodes are fixed when the partition is set / the table is uploaded: a node that moves
    // stays with its rank (no other rank could learn that it left).
    int sp_part = 0, sp_parts = 0;
    std::vector<int32_t> sp_nodes;   // the region's nodes, ascending
    DevBuf<uint8_t> d_member;        // [n] 1 = a receiver of this context (spatial partitions; the reception stage's "owned")
    DevBuf<int32_t> d_draw_nodes;    // spatial partitions whose links draw: node index of every drawing link, packet-major
    DevBuf<uint32_t> d_all_off;      // [world][n_new] scratch of rm_tick_finish_draws_nodes
    DevBuf<int32_t> d_all_nodes;     // its host lists, uploaded
    // RCCL inside the library (rm_comm_*, rm_dist_*; rm_api_comm.cpp): this context's rank of a communicator
    void *comm = nullptr;            // ncclComm_t
    bool comm_owned = false;
    int comm_world = 1, comm_rank = 0;
    DevBuf<rm_tx_record> d_dist_mine, d_dist_all; // this rank's packed frames / the frames of all ranks [rank][tick][slot]
    DevBuf<int32_t> d_dist_idx;                   // the gathered source indices [rank][tick][slot] (what crosses the links)
    uint32_t cap = 1u << 22;
    int last_tile_reuse = 1; // ticks of the last batch a filter workgroup swept with one load of its receivers (rm_batch_tile_reuse)
    // Digest of the node table (rm_table_digest): xor over the nodes of a 64-bit hash of (index, every field), mixed with the node
    // count -- a function of the table's CONTENT, whatever sequence of uploads and updates produced it.  Ranks that exchange
    // source indices build each other's records from their own copies of the table: the digests ride in the all-gather and a
    // rank whose copy differs is found out (rm_dist_batch_run_sources_device, rm_batch_run_gathered_blocks_device).
    uint64_t table_xor = 0, table_digest = 0;
    DevBuf<int32_t> d_dist_stage; // this rank's block of a sharded batch: its source indices, then the trailer with the digest

    int64_t current_time = 0;
    int64_t t_begin = 0, t_end = 0;
    bool in_tick = false;

    // on-air list (host-record mode)
    std::vector<rm_tx_record> pending; // frames enqueued in the current tick
    // on-air list (device-source mode, SINR): the live batches are a window [air_head, air_tail) of
    // d_air; a batch = the frames of one rm_tick_run_sources_device call (same start and air time)
    struct AirBatch {
        int count;
        int64_t end_us;
        uint32_t tick; // AirLists::tick of the call that put the batch on the air
    };
    // the window holds frames that were selected for this partition's region (k_rank_frames over a batch of overlapping SINR ticks:
    // rm::CullEntry): the boxes the selections were made against, and until when each matters
    bool air_culled = false;
    DevBuf<rm::CullEntry> d_cull_ring;
    int64_t cull_end[rm::kCullRing] = {};
    uint32_t cull_seq = 0;
    DevBuf<rm_tx_record> d_air, d_air_alt; // (the window's buffer and the one it slides into when this one is used up: air_window_reserve)
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
        uint64_t scans = 0;       // ticks evaluated by scan (rm_airscan.hip): they leave nothing in the lists
        uint32_t stamp = 0;       // of the nodes' chains of own frames (TickSlot::d_self_slot of the context): one per tick by scan / per batch of overlapping ticks
    } air;
    int64_t air_max_t_begin = INT64_MIN; // the latest t_begin a tick over the on-air window has had (air_tick_device)
    // a batch of SINR ticks whose frames outlive their tick (rm_airbatch.hip, rm_api_airbatch.cpp): the batch's index of
    // the frames it can see, the surviving (link, frame) pairs, the descriptors' pinned staging
    struct Overlap {
        DevBuf<float4> fr_f, e_f;
        DevBuf<int4> fr_m, e_m;
        DevBuf<longlong2> fr_t, e_t;
        DevBuf<uint32_t> fr_bin, bin_cnt, bin_off, block_sum, every, misc, pair_tail, items;
        DevBuf<int32_t> self_next, slot_first;
        DevBuf<uint8_t> defer;
        DevBuf<rm::OvTick> ticks;
        DevBuf<rm::OvPair> pairs;
        size_t pair_cap = 0;        // entries over all shards
        size_t pair_cap_forced = 0; // RM_OV_PAIR_CAP (tests)
        char *h_desc[2] = {nullptr, nullptr}; // pinned: OvTick[RM_MAX_BATCH] then slot_first
        size_t h_desc_bytes[2] = {0, 0};
        hipEvent_t h_ev[2] = {nullptr, nullptr};
        int gen = 0;
        uint32_t *h_flag = nullptr; // pinned, one word per generation: OvDev::misc[1] of the batch that used the generation last (frames
                                    // were deferred, the pair list was full: grown for the next batch of that generation); read only
                                    // after that batch's copy has landed (h_flag_ev)
        hipEvent_t h_flag_ev[2] = {nullptr, nullptr};
        bool h_flag_used[2] = {false, false};
        uint64_t batches = 0, ticks_done = 0, last_frames = 0;
    } ov;
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
    const rm_tx_record *host_src = nullptr; // the tick being prepared reads its records from this host-mapped block (prepare_tick)
    uint32_t transmit_seq = 0;
    DevBuf<rm::TickDev> d_ticks; // [RM_MAX_BATCH] descriptors of the running rm_batch_* call
    DevBuf<int32_t> d_near_list;  // [ticks][blocks of kNearSb filter workgroups][frames] near-frame lists of a batch over a large table
    DevBuf<uint32_t> d_near_cnt;  // [ticks][blocks]
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

    // profiling (rm_profile_enable): on every n-th launch sequence each kernel launch carries its own pair of events
    // (rm::KernelProbe); the intervals are summed per stage and per kernel
    bool profile = false;   // sampling on
    int profile_every = 1;  // sample every n-th launch sequence
    uint64_t tick_index = 0;
    struct Sample {
        struct K {
            hipEvent_t a = nullptr, b = nullptr;
            int stage = 0;
            const char *name = nullptr;
        };
        std::vector<K> k;   // (events are created once and reused)
        int n = 0;          // launches probed in this sample
        int cur_stage = 0;  // stage the launches being issued belong to
    };
    std::deque<Sample> ev_pool;
    size_t ev_used = 0;
    uint32_t prof_samples = 0;
    double prof_ms[RM_PROFILE_STAGES] = {0};
    struct KernelTime {
        std::string name;
        int stage;
        uint32_t launches;
        double ms;
    };
    std::vector<KernelTime> prof_kernels;
};

namespace rmh {

// ---- rm_api_context.cpp: model, derived constants
const char *model_name(int kind);
bool is_sinr(const rm_context *c);
int part_first(const rm_context *c); // first node index the position map covers
int part_count(const rm_context *c); // receivers of this context
int pos_span(const rm_context *c);   // node indices the position map covers (index partition: its nodes; spatial: all)
bool part_spatial(const rm_context *c);
bool frac(double p);
bool maybe_draws(rm_context *c);
int validate_model(const rm_model_params *p);
void recompute_frame(rm_context *c);
rm::ModelDev model_dev(const rm_context *c);
rm::NodesDev nodes_dev(rm_context *c);
bool is_geometric(const rm_context *c);
int build_shadow_table(rm_context *c);

// ---- rm_api_nodes.cpp: the receiver table
int rebuild_receivers(rm_context *c);
int select_region(rm_context *c);    // the member nodes of a spatial partition from the current node table
int patch_nodes(rm_context *c, const int32_t *nodes, int count);
int prepare_nodes(rm_context *c);

template <typename T> int upload(DevBuf<T> &d, const std::vector<T> &h, hipStream_t s)
{
    RM_HIP(d.ensure(std::max<size_t>(h.size(), 1)));
    if (!h.empty()) RM_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return RM_OK;
}

// ---- rm_api_plan.cpp: one tick's buffers, descriptor and launch sequence
// Link-sized buffers of a result slot.  Only what the configuration touches is allocated (a
// batch keeps up to RM_MAX_BATCH slots): `payload` = per-entry rssi / probability / node index
// (unsorted tables and SINR), `sinr` = per-receiver lists and linear powers, `draws` = the
// java.util.Random scan and the probabilities carried to it.
enum { kFeatPayload = 1, kFeatSinr = 2, kFeatDraws = 4 };
int ensure_link_buffers(rm_context *c, TickSlot &ts, int feat);

// What one tick's launch sequence needs besides the slot: filled by prepare_tick.
struct TickPlan {
    rm::TickDev t{};
    rm::ScanDev scan{};  // (t.air_scan only)
    rm::LaunchCfg cfg{};
    bool sinr = false, stochastic = false, partitioned = false;
    bool empty = false; // nothing to sweep: the result is an empty one
};

// Buffers and descriptor of one tick in result slot `ts`; `tx` is the on-air list in device memory
// (build mode: where the records of the source indices `src_list` are written).
// kAirScan: `tx` holds every frame on the air (as for a rebuild), only the new ones are evaluated, their interferers are found
// among the frames themselves (rm_airscan.hip); the lists are not touched and count as stale afterwards
// kAirBatch: a tick of a batch of overlapping SINR ticks (rm_api_airbatch.cpp): swept as the medium without SINR (heard links
// only, cut-off at the sensitivity); the interference stages of the whole batch follow the sweep
enum { kAirNone = 0, kAirIncremental = 1, kAirRebuild = 2, kAirScan = 3, kAirBatch = 4 };
bool air_scan_applies(rm_context *c, int n_new); // (after prepare_nodes)
uint32_t air_sub_cap(const rm_context *c);
bool air_lists_current(const rm_context *c, int64_t t_begin, uint32_t oldest);
int prepare_tick(rm_context *c, TickSlot &ts, TickPlan &plan, bool want_wg, const rm_tx_record *tx, int n_active,
                 int first_new, const int32_t *src_list = nullptr, int64_t src_start_us = 0, int64_t src_air_us = 0,
                 int air_mode = kAirNone, uint32_t air_oldest = 0, const rm::PlanKnobs *knobs_in = nullptr);
int launch_tick(rm_context *c, TickSlot &ts, const TickPlan &plan);
int materialize(rm_context *c, TickSlot &ts);
int dense_layout(rm_context *c, TickSlot &ts); // (the cells' offsets and totals of a dense tick that ended with its cells)
int run_tick(rm_context *c, const rm_tx_record *tx, int n_active, int first_new, const int32_t *src_list = nullptr,
             int64_t src_start_us = 0, int64_t src_air_us = 0, int air_mode = kAirNone, uint32_t air_oldest = 0, bool ev_may_wait = false);
int drain_profile(rm_context *c);
// a sampled launch sequence: begin_sample() returns the sample (or nullptr: not sampled) and routes the kernel probes of
// this thread to it; sample_stage() names the stage of the launches that follow; end_sample() unroutes.  (ProbeScope
// does the last two on every way out of a launch function.)
rm_context::Sample *begin_sample(rm_context *c);
inline void sample_stage(rm_context::Sample *smp, int stage) { if (smp) smp->cur_stage = stage; }
void end_sample();
struct ProbeScope {
    rm_context::Sample *smp;
    explicit ProbeScope(rm_context *c) : smp(begin_sample(c)) {}
    ~ProbeScope() { if (smp) end_sample(); }
    ProbeScope(const ProbeScope &) = delete;
    ProbeScope &operator=(const ProbeScope &) = delete;
};

// ---- rm_api_events.cpp: the reception stage
rm::EvDev ev_dev(rm_context *c);
int ev_ensure_nodes(rm_context *c);
int ev_append(rm_context *c, TickSlot &ts, bool may_wait = false);
int ev_flush_append(rm_context *c);

// ---- rm_api_tick.cpp: results of an evaluated tick, the host-mapped result block
rm_tx_record make_record(const rm_context *c, int32_t src, int64_t start_us, int64_t air_us, const double *txpower,
                         const int32_t *channel);
const char *record_flag_message(uint32_t flag);
int copy_out(rm_context *c, TickSlot &ts, int32_t *pkt, int32_t *dst, uint8_t *verdict, double *rssi, double *sinr,
             uint32_t cap, uint32_t *count, uint8_t *pkt_interference, uint32_t *pkt_offset);
size_t pad64(size_t v);
rm::HostView stage_view(char *base, uint32_t links, uint32_t packets, size_t *bytes);
rm::BatchCounts *stage_counts(char *base);
int ensure_stage(rm_context *c, uint32_t links, uint32_t packets);
int pack_to_stage(rm_context *c, TickSlot &ts, rm::HostView *view);
// does this slot's result go to the host with ONE rssi per packet?  The reference's media hand the packet's transmit power to
// every heard link unchanged; the log-distance medium computes a link's own (RM_HOST_LINK_RSSI=1 keeps the column for all)
bool host_pkt_rssi(const rm_context *c, const TickSlot &ts);
int stage_status(rm_context *c, const rm::HostView &v);
int tick_run_host(rm_context *c);
// The SINR medium's tick: the frames of earlier ticks that are still on the air stay resident on the device (the window
// [air_head, air_tail) of d_air); the new ones are built from source indices (dev_src), or given as records in device memory
// or -- new_on_host -- in the host's pinned staging block, and join the window's tail.
int air_tick_device(rm_context *c, int64_t t_begin_us, const int32_t *dev_src, const rm_tx_record *dev_new, int32_t n, int64_t start_us,
                    int64_t air_us, int64_t latest_end_us, bool new_on_host);
int air_window_expire(rm_context *c, int64_t t_begin_us);
int air_window_reserve(rm_context *c, size_t n_more);
int result_device(rm_context *c, TickSlot &ts, rm_device_result *out);
int result_count(rm_context *c, TickSlot &ts, uint32_t *count, uint32_t *dropped);

// ---- rm_api_comm.cpp
int comm_all_gather(rm_context *c, const void *mine, void *all, size_t bytes);
int comm_finish_draws(rm_context *c);
int group_comm_init(rm_context *const *members, int n, void **comms_out);
int group_all_gather(rm_context *const *members, int n, const void *const *mine, void *const *all, size_t bytes);

// ---- rm_api_batch.cpp
TickSlot *slot_of(rm_context *c, int32_t slot);
// the launch sequence of n prepared ticks; m_override: the model the sweep runs with (a batch of overlapping SINR ticks
// sweeps with the medium without SINR), after_sweep: issued behind the last sweep stage, inside the sampled sequence
int launch_batch(rm_context *c, TickSlot *const *slots, const TickPlan *plans, int n, const rm::ModelDev *m_override = nullptr,
                 int (*after_sweep)(rm_context *, void *) = nullptr, void *after_arg = nullptr, const rm::RankFramesArgs *rank_frames = nullptr);
// a rank's frame list for tick `t` of a batch of gathered source indices (plan and slot prepared; n_pub gathered slots)
bool rank_frames_wanted(rm_context *c);
int plan_rank_frames(rm_context *c, TickSlot &ts, rm::TickDev &t, int n_pub);
// ---- rm_api_airbatch.cpp
bool overlap_wanted(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int32_t *n_per, const int64_t *start_us,
                    const int64_t *air_us);
int batch_run_overlap(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us, const int32_t *const *dev_src,
                      const int32_t *n_per, const int64_t *start_us, const int64_t *air_us, const rm_tx_record *gathered, int gather_world,
                      int gather_slots, const int32_t *gathered_idx, int gather_block = 0, int digest_off = -1);
// gathered / gathered_idx: the ticks' frames where an all-gather of per-rank blocks left them, [rank][tick][slot] -- as records, or
// as source indices (then start_us / air_us give the ticks' time spans and every record is built from the node table)
int batch_run(rm_context *c, int32_t n_ticks, const int64_t *t_begin_us, const int64_t *t_end_us, const int32_t *const *dev_src,
              const rm_tx_record *const *dev_new, const int32_t *n_per, const int64_t *start_us, const int64_t *air_us,
              const rm_tx_record *gathered = nullptr, int gather_world = 0, int gather_slots = 0, const int32_t *gathered_idx = nullptr,
              int gather_block = 0, int digest_off = -1);
// (gather_block: elements from one rank's block of the gathered buffer to the next, 0 = n_ticks * gather_slots; digest_off: where
// in a rank's block of source indices its node-table digest lies -- two words, rm_table_digest -- or -1)

} // namespace rmh
