// rm_engine.h -- internal interface between the C-ABI host code (rm_api_*.cpp behind rm_host.hpp) and the
// gfx950 kernels (rm_*.hip).  Not part of the public boundary (include/radiomedium_hip.h).
//
// The gfx950 (CDNA4, wave64) kernels of the radio-medium engine, in short:
//
// k_filter sweeps every (frame on the air) x (receiver of this rank's partition) link: one
// receiver per lane (RPT groups of 64 per wave, resident in registers), transmitter tiles of 64
// frames staged in LDS, a bounding-box test of the tile against each spatially sorted receiver
// group (one frame per lane, one ballot), then a conservative fp32 (or fp64) geometric pre-filter
// on the near frames whose ballots append candidate links to a compact list (one atomic per wave
// step).  k_exact evaluates the candidates with full lanes: the reference's fp64 arithmetic in the
// reference's operation order.  Everything after that is O(heard links): offsets from the
// (frame, slab) cell counts, SINR over per-receiver lists, ordered scatter, per-packet reorder to
// node-index order, Java-RNG draws.
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (no fast-math): the exact path relies
// on every fp64 operation being one IEEE-754 rounding, as in Java.
//
// Reference paths: /root/reference/radio-medium/java/se/sics/emul8/radiomedium/.
//
// Files: rm_math.hpp (exact arithmetic: E-math, link hash, Q80, java.util.Random), rm_device.hpp
// (shared device code: wave helpers, pre-filter records, eval_link, fused scans), rm_filter.hip,
// rm_exact.hip, rm_reorder.hip, rm_transmit.hip, rm_tick.hip, rm_events.hip (kernels + their launchers),
// rm_evorder.hpp (the reference event queue's pop order as a sort key), rm_api_*.cpp (C ABI, by concern).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/radiomedium_hip.h"

namespace rm {

// Kernel probes (rm_profile_enable).  On a sampled launch sequence every kernel launch binds a pair of events to its
// OWN dispatch (hipExtLaunchKernelGGL: the events take the start and the end of that kernel -- the interval rocprofv3's
// kernel trace reports -- not the stream time around it, which also holds the launch gap and whatever other contexts
// have in flight).  Unsampled sequences launch as ever.
struct KernelProbe {
    bool (*take)(void *user, const char *kernel, hipEvent_t *start, hipEvent_t *stop);
    void *user;
};
extern thread_local KernelProbe g_probe;
#define RM_KLAUNCH(kern, grid, block, shm, s, ...)                                                      \
    do {                                                                                                \
        hipEvent_t pa_ = nullptr, pb_ = nullptr;                                                        \
        if (rm::g_probe.take && rm::g_probe.take(rm::g_probe.user, #kern, &pa_, &pb_))                  \
            hipExtLaunchKernelGGL(kern, grid, block, shm, s, pa_, pb_, 0, __VA_ARGS__);                 \
        else                                                                                            \
            hipLaunchKernelGGL(kern, grid, block, shm, s, __VA_ARGS__);                                 \
    } while (0)

constexpr int kTxChunk = 64;        // transmitters per LDS tile == wave width (one lane per frame)
constexpr int kWavesPerBlock = 4;   // 256-thread workgroups
constexpr int kBlock = 64 * kWavesPerBlock;
constexpr int kGroup = 64;          // receivers per bounding-box group (one per lane)
constexpr int kShards = 256;        // append counters of the candidate list (one word sustains only ~88 atomics/us)
constexpr int kShadowBins = 256;    // bins over rho = d^2 / cut^2 of the shadowing table
constexpr double kShadowPad = 0.02; // relative pad on rho folded into the table
enum { kFilterGrid = 0, kFilterWg = 2 }; // TickDev::filter_mode (plan_filter)
constexpr int kShardStride = 32;
constexpr int kSgG = 64, kSgCells = kSgG * kSgG, kSgK = 16, kSgMax = 64; // the frame grid of a tick by scan (TickDev::sg_*)
constexpr int kFusedScanMax = 8192; // per-frame counts a consumer kernel scans by itself in LDS (block_scan_counts, rm_device.hpp)
constexpr int kFrameSegMax = 512;   // heard links of one frame kept in LDS by k_tick_frames (rm_tick.hip) = its fixed segment of the A records
constexpr int kMaxBatch = RM_MAX_BATCH; // ticks per batched launch (descriptors in device memory)    // u32 words between shard counters: one 128-byte line each

// link-entry flags
constexpr uint8_t kFlagHeardNew = 1;   // gets an output record
constexpr uint8_t kFlagInterferer = 2; // rssi >= interference floor (SINR mode)
constexpr uint8_t kFlagSelf = 4;       // receiver is the source of this on-air frame (half duplex)
constexpr uint8_t kFlagTxDead = 8;     // the frame's txSuccess is <= 0: every heard link is interfered, no draw

// model constants as the kernels need them
struct ModelDev {
    int kind;
    int flags;
    double udgm_ratio_rx;     // successRatioRx
    double udgm_range;        // transmissionRange
    double const_range;
    const double *n2n;        // row-major m x m (device)
    int n2n_m;
    double ld_pl0, ld_exp, ld_d0, ld_sigma, ld_clip;
    uint64_t ld_seed_mixed;   // mix64(seed + golden)
    double ld_sens, ld_noise, ld_capture, ld_ifloor;
    double ld_noise_lin;      // det_pow10(noise/10)
    double ld_level;          // candidate level L = sens, or min(sens, ifloor) with SINR
    double ld_cut_scale;      // log2(10) / (10 n): margin [dB] -> log2 of the cut-off distance ratio (pre-filter only)
    // pre-filter
    double org_x, org_y, org_z; // origin of the fp32 frame
    double coord_bound;         // max |coord - origin| the fp32 slack was computed for
    double f32_slack;           // Delta (metres) added to every cut-off distance
    double geo_cut;             // cut-off distance for UDGM / CONST (metres), <0 = nobody
    const uint32_t *shadow_tbl; // [kShadowBins] second-level filter of the shadowed log-distance medium (or null)
};

// Node state resident in HBM.
//  * source table, indexed by node index (registration order): what a RadioPacket copies from
//    its source (RadioPacket.java:46-52) -- used to build Tx records on the device;
//  * receiver table of this rank's partition, in ENGINE ORDER: receivers are spatially sorted
//    (k-d split down to groups of 64) so that a wave's 64 receivers share a small bounding box;
//    orig[pos] maps back to the node index.
// exact-path view of one receiver: one aligned 64-byte record, one gather per candidate link
struct alignas(64) RxRecord {
    double x, y, z;   // Position.java:37-39
    double rxprob;    // Transciever.java:17
    int32_t orig;     // node index (registration order)
    int32_t int_id;   // Node.getIdAsInteger()
    int32_t channel;
    int32_t enabled;
    double pad[2];
};

// One changed node (node-config-set, SimulatorJSONHandler.java:102-143): its new state for the source
// table and, when it is a receiver of this partition with a current table, its engine position.
struct NodePatch {
    int32_t node;     // node index
    int32_t pos;      // engine position in the receiver table, -1 = not a receiver here / table stale
    double x, y, z, txpower, txprob, rxprob;
    int32_t channel, enabled;
};

// The same receiver for the exact stage of the spatially sorted media, in 32 bytes: every candidate of
// those media passed the sweep's exact channel test and its "radio enabled" test (a disabled radio has NaN
// pre-filter coordinates), so the record carries neither; a reception probability other than 1.0 is
// flagged and read from the rxprob column.  Half the bytes per gathered candidate, and the table of
// 100 k receivers (3.2 MB) fits one XCD's L2.
struct alignas(32) RxCompact {
    double x, y, z;
    int32_t orig;     // node index (registration order)
    uint32_t flags;   // bit 0: rxprob != 1.0
};

// What a RadioPacket copies from its source (RadioPacket.java:46-52), one aligned 64-byte record per node: a frame named by its
// source index is ONE gather (the columns below cost six lines per frame; at a million nodes each of them left the L2).
struct alignas(64) SrcRecord {
    double x, y, z, txpower, txprob;
    int32_t channel, int_id;
    double pad[2];
};
static_assert(sizeof(SrcRecord) == 64, "one cache line");

struct NodesDev {
    int n;                                   // nodes in the simulator
    const SrcRecord *srec;                   // [n] the source table by node index, as records
    const double *sx, *sy, *sz, *stxpower, *stxprob;
    const double *srxprob;                   // Transciever.rxProbability by node index (the dense tick visits receivers in node order)
    const int32_t *schannel, *sint_id;
    const uint8_t *senabled;                 // Transciever.isEnabled by node index (node-info)
    int n_rx;                                // receivers of this partition
    const double *x, *y, *z, *rxprob;
    const int32_t *channel, *int_id, *orig;
    const uint8_t *enabled;
    const RxRecord *rec;                     // [n_rx] the same data as one record per receiver (k_exact, unsorted tables; k_transmit_one)
    const RxCompact *rec32;                  // [n_rx] 32-byte form for the exact stage of sorted tables
    const int32_t *pos_of;                   // [pos_span] node index - rx_first -> engine position, -1 = not a receiver here
    int rx_first;                            // index partition: its first node, pos_span = n_rx; spatial partition: 0, pos_span = n
    int pos_span;
    float4 *rxf;                             // pre-filter record: (fx, fy, fz, channel bits); NaN = never a candidate
    float4 *bbox_xy;                         // per group of 64: (minx, miny, maxx, maxy) in the fp32 frame
    float2 *bbox_z;                          //                 (minz, maxz)
    float4 *wg_box_xy;                       // per filter workgroup (16 groups = 1024 receivers): union of its boxes
    float2 *wg_box_z;
    uint32_t *grp_chmask;                    // per group of 64: bit (channel & 31) of every receiver with its radio on -- a frame on a channel
    uint32_t *wg_chmask;                     // whose bit is not set cannot be heard in the group (per filter workgroup: the OR of its groups)
};

// SINR across ticks: the frames on the air leave their significant links in per-receiver lists that live on the
// device from tick to tick, so that a tick evaluates its NEW frames only.  An entry carries everything the interference
// sum reads.  Entries are allocated from kShards sub-rings in tick order; a tick's allocations begin at
// mark[tick][sub], and everything older than the oldest tick that still has a frame on the air (wtick) is free again.
// Links are validated while walking: a head or a pointer is followed only while the ticks do not increase along the
// list and stay >= wtick (an insertion stores the previous head only if it was live, so a reclaimed slot can only come
// back with a newer tick than the entry that points to it).
// One 32-byte entry = half a cache line, never straddling two.  (What a walk costs is the number of cache lines a
// compute unit has to fetch at random, not the length of the chain: entries with four successor pointers, and copies of a
// receiver's newest entries side by side, were both measured and made the walk no faster, the insertion slower.)
struct alignas(32) AirEntry {
    int64_t start_us;
    double lin;       // linear power at the receiver
    uint32_t air_us;  // SINR frames are shorter than 2^32 us (checked when they are enqueued)
    int32_t next;
    uint32_t meta;    // tick << 2 | kAirSelf | kAirInterferer
    uint32_t pad;
};
static_assert(sizeof(AirEntry) == 32, "half a cache line");
constexpr uint32_t kAirSelf = 1, kAirInterferer = 2;
constexpr uint32_t kAirTicks = 1024;      // marks kept: a frame that stays longer makes the host rebuild the lists
constexpr uint32_t kAirTickMax = (1u << 30) - 8;

struct AirDev {
    AirEntry *pool;            // nullptr: the lists are per tick (st_next / head; batches of self-contained ticks)
    unsigned long long *head;  // [n_rx] tick << 32 | entry; 0 = empty (ticks start at 1)
    uint32_t *tail;            // [kShards * kShardStride] entries ever allocated in the sub-ring
    uint32_t *mark;            // [kAirTicks][kShards] tail when the tick began
    uint32_t *bad;             // [1] an allocation ran over live entries: the lists are unusable until rebuilt
    uint32_t sub_mask, sub_shift;
    uint32_t tick, wtick;
    int64_t t_begin;           // entries with start + air <= t_begin have left the air
};

// The SINR medium's lone tick "by scan" (rm_airscan.hip): every heard link of a new frame finds its interferers among the
// frames on the air themselves -- no per-receiver lists.  The first launch (k_tick_frames) leaves one scan record per
// frame on the air and indexes them, from scratch every tick (nothing to keep current): a kSgG x kSgG grid over the fp32
// frame with up to kSgK frames per cell, the rest -- and frames without a place or a bound -- in a list every new frame
// looks at; the largest radius; and per NODE the chain of its own frames on the air (half duplex).
struct ScanDev {
    double level;           // the level a link has to reach to interfere (ModelDev::ld_level of the SINR medium)
    float4 *xyzr;           // [n_active] position in the fp32 frame + cut-off radius at that level (-1: nobody; +inf: no bound)
    int32_t *ch;            // [n_active] channel
    uint32_t *cnt;          // [kSgCells] frames per cell, [kSgCells] entries of `every`, [kSgCells + 1 ..][kSgMax] largest radius (float bits)
    uint32_t *cnt_next;     // the other parity, zeroed by k_sinr_scan for the next tick by scan
    float4 *bucket_xyzr;    // [kSgCells][kSgK] the cell's frames: their scan records (a new frame tests them without a second round trip) ...
    int2 *bucket_ci;        // ... channel, frame index
    uint32_t *every;        // [n_active]
    unsigned long long *self_slot; // [n] stamp << 32 | newest frame of the node on the air
    int32_t *self_next;     // [n_active] the node's frame before that one, -1: none
    uint32_t stamp;         // this tick's stamp (never 0)
    float half, inv;        // cell = int((x + half) * inv)
};

// A BATCH of SINR ticks whose frames outlive their tick (rm_airbatch.hip; BASELINE configs[4]).  The frames the batch can
// see -- the window of frames still on the air from earlier calls, then the batch's ticks one after the other -- are ONE
// array, cut into time slots (a window batch or a tick each).  The heard links of all ticks come from the sweep of the medium
// without SINR; then every frame is indexed once per batch by (cell of the fp32 frame's grid, slot), slot fastest, so that
// "the frames near this place that can still be on the air in tick b" -- slots slot_lo(b) .. slot(b) -- is ONE contiguous
// run per cell; a new frame looks only at frames of its own and earlier slots (verdicts are causal: DESIGN.md section 6, E4).
struct OvTick {                 // one tick of the batch as the interference stages see it
    int frame_first, n_new;     // its new frames in OvDev::tx
    int slot, slot_lo;          // its time slot; the oldest slot that may hold a frame still on the air when it begins
    int64_t t_begin;
    int shift, pad;             // TickDev::shift of its result slot
    const uint32_t *slot_off;   // its heard links, packet-major (the compact arrays of the tick's result slot) ...
    const int32_t *out_dst;
    const double *out_rssi;
    double *out_sinr;           // ... and what the stages write
    uint8_t *out_verdict;
    unsigned long long *acc_lo, *acc_hi; // [links] Q80 interference sum per heard link
    uint8_t *hd;                // [links] half duplex: the receiver is itself on the air
    uint32_t *flags;            // the tick's TickDev::stage_count ([1]: dropped for capacity)
};
struct OvPair {                 // a (heard link, frame on the air) pair that passed the conservative tests
    uint32_t link, tick;
    int32_t pos, frame;         // the link's receiver (engine position), the frame (index in OvDev::tx)
};
struct OvDev {
    const rm_tx_record *tx;     // [n_frames]
    int n_frames, n_slots, n_ticks, n_bins; // n_bins = kSgCells * n_slots
    int max_new;                // most new frames of a tick of the batch
    const int32_t *slot_first;  // [n_slots + 1] first frame of a slot
    const OvTick *ticks;        // [n_ticks]
    // per frame: pre-filter record at the interference level (position in the fp32 frame, threshold; w < 0: reaches nobody),
    // (shadow-bin scale, source node, channel, frame index), (start, end), bin
    float4 *fr_f;
    int4 *fr_m;
    longlong2 *fr_t;
    uint32_t *fr_bin;           // 0xFFFFFFFF: not in the grid
    uint32_t *bin_cnt, *bin_off, *block_sum; // [n_bins], [n_bins + 1], [ceil(n_bins / kOvScanBlock)]
    float4 *e_f;                // the same records sorted by bin
    int4 *e_m;
    longlong2 *e_t;
    uint8_t *defer;             // [n_frames] 1: the frame's pairs did not all fit the pair list -- its sums are formed by the second go
    uint32_t *every;            // [n_frames] frames without a cell (no bound, outside the frame)
    uint32_t *misc;             // [0] entries of `every`, [1] a frame was deferred (the pair list was full), [2] pairs that interfered (statistics), [3] items of the pair stage, [8 .. 8 + kSgMax) largest radius (float bits)
    uint32_t *items;            // [sum of the ticks' new frames] tick << 16 | frame, tick-major (k_ov_count)
    unsigned long long *self_slot; // [n] stamp << 32 | newest frame of the node
    int32_t *self_next;         // [n_frames]
    uint32_t stamp;
    float half, inv;
    uint32_t *h_flag;           // the host's word (pinned) that takes misc[1] at the batch's end
    OvPair *pairs;              // kShards regions of pair_seg entries
    uint32_t *pair_tail;        // [kShards * kShardStride]
    uint32_t pair_seg;
};
constexpr int kNearSb = 16;   // filter workgroups (of 1024 receivers) per block of the near-frame lists
constexpr int kOvScanBlock = 4096;

struct TickDev {
    AirDev air;
    const rm_tx_record *tx; // on-air list, canonical order [n_active]
    // build mode (single tick of new frames given as source indices): k_filter builds the records
    // from the source table while staging its tile and writes them to tx_build (== tx) for k_exact
    const int32_t *src_list;
    rm_tx_record *tx_build;
    int64_t src_start_us, src_air_us;
    // gather mode (a tick of a receiver-sharded batch): the frames are read from the buffer an all-gather of per-rank blocks
    // [rank][tick][slot] left them in -- frame i of this tick is gather_src[(i / gather_slots) * gather_stride + i % gather_slots]
    // -- and copied to tx_build (== tx) by k_tick_prep, where every later stage finds them in the tick's packet order
    const rm_tx_record *gather_src;
    const int32_t *gather_idx; // ... or, instead of records, the frames' SOURCE INDICES in the same layout (what the all-gather of a
                               // sharded batch carries): k_tick_prep builds the record from the node table (src_start_us / src_air_us)
    int gather_slots, gather_stride;
    // A rank's frame list (k_rank_frames; gathered source indices over a receiver PARTITION): of the world * slots frames of the tick
    // only those whose reach touches the partition's receivers are kept: their records and pre-filter records are written in
    // place, n_active = their number -- written into the descriptor ON THE DEVICE -- and
    // fl_map (local frame -> the frame's number among the gathered slots: packets keep their global numbers in the results);
    // fl_lb[g] = listed frames before gathered slot g ([n_pub + 1]); pub_off[n_pub + 1]: the packets' offsets by GLOBAL number,
    // written by the reorder stage's publisher (slot_off stays local).  n_pub = world * slots (0: no list).
    int32_t *fl_map;
    uint32_t *fl_lb;
    uint32_t *pub_off;
    int n_pub;
    int fl_pad;             // the tick's records are part of the on-air window: the slots behind the listed frames get padding records
    int32_t *fl_ov_n_new;   // (a batch of overlapping SINR ticks) OvTick::n_new of this tick, in device memory: patched as n_active is
    int n_active;
    int first_new;          // frames [first_new, n_active) get verdicts
    int first_eval;         // frames [first_eval, n_active) are swept by the filter kernel
    int cnt_base;           // eval-relative index of the first counted slot (multiple of 64)
    int shift;              // slot of new packet 0
    int n_cnt;              // counted slots (multiple of 64)
    int n_rx;               // receivers of the partition
    int rpt;                // receiver groups per wave in the filter kernel
    int n_slabs;            // ceil(n_rx / (64*rpt))
    int filter_mode;        // kFilterGrid / kFilterWg
    int reset_heads;        // k_tick_prep empties the per-receiver link lists (SINR ticks of a batch)
    int seg_ordered;        // the frames' segments are already in node order (rm_tick.hip): k_reorder only compacts them
    int check_txprob;       // records given by the caller on the device, tick evaluated without the draw kernels: a record whose
                            // txprob is strictly between 0 and 1 would need a draw -- flagged (RM_ERR_STATE when the result is read)
    int air_scan;           // the SINR medium's lone tick "by scan" (rm_airscan.hip; ScanDev below): `tx` holds EVERY frame on the
                            // air, only the new ones are swept (first_eval == first_new), nothing is kept per receiver
    int check_span;         // k_tick_prep verifies that every frame lies inside [span_begin, span_end] (SINR ticks whose
    int64_t span_begin, span_end; // records come from the caller: the batch is only valid if its ticks are self-contained)
    // per (slot, slab) heard counts / offsets (off is relative to the frame's first link), layout [(chunk*n_slabs + slab)*64 + lane]
    uint32_t *cnt, *off;
    uint32_t *slot_tot;     // [n_cnt] heard links per frame slot
    uint32_t *slot_off;     // [n_cnt + 1] exclusive scan of slot_tot
    // candidate / link list (unordered; appended by the filter kernel, completed by the exact kernel)
    uint32_t *stage_count;  // [1] dropped-for-capacity flag
    uint32_t *next_counters; // the other parity's 8 counters, zeroed for the next tick by k_filter
    uint32_t *shard_count;  // [kShards * kShardStride] entries appended per shard
    uint32_t *next_shard_count;
    uint32_t cap;           // capacity of the ordered output records
    uint32_t seg_cap;       // candidate entries per shard (shard s owns [s*seg_cap, (s+1)*seg_cap))
    uint32_t shard_mask;    // shards in use - 1: all kShards for a lone tick; a batch appends fewer, longer runs and uses
                            // 64, so that the exact kernel's 256-entry chunks are full (195 entries per shard fill 76 %)
    int use_matrix;         // 1: ordered scatter through the (frame, slab) cell matrix (unsorted table)
                            // 0: per-frame counts + cursor, order restored by k_reorder (sorted table)
    uint32_t *cursor;       // [n_cnt] per-frame scatter cursor = heard links of the frame (use_matrix == 0)
    uint32_t *cand_tot;     // [n_cnt] candidate links per frame, this tick's parity (use_matrix == 0)
    uint32_t *cand_tot_next; // the other parity, zeroed by k_filter for the next tick
    uint32_t *seg_off;      // [n_cnt + 1] exclusive scan of cand_tot: the frame's segment in the A records
    int zero_len;           // slots of cursor / cand_tot_next that k_filter has to zero
    // two-level filter (k_tick_prep + k_filter_wg): per-frame pre-filter records
    float4 *p_txf;          // [n_eval]
    int32_t *p_ch, *p_src;  // [n_eval]
    float *p_inv;           // [n_eval]
    // batches over large tables: per block of kNearSb filter workgroups (16 k receivers) the frames near its box, found by a
    // pre-pass (k_near_lists) -- phase A of a workgroup then looks at these few dozen instead of at every frame of the tick
    const int32_t *near_list; // [blocks][near_cap] frame numbers (nullptr: phase A looks at all frames)
    const uint32_t *near_cnt; // [blocks]
    int near_cap;
    int32_t *a_e;           // [..] link-entry index of an A record (SINR results are looked up through it)
    int32_t *st_pkt;        // eval-relative frame index
    int32_t *st_dst;        // receiver engine position
    uint32_t *st_blk;       // index of the first entry of this entry's (frame, slab) block
    double *st_aux;         // rssi as reported (packet txpower for the reference media)
    double *st_prob;        // receive probability of the link
    int32_t *st_orig;       // receiver node index
    double *st_lin;         // linear power (SINR)
    double *st_sinr;
    int32_t *st_next;       // per-receiver list (SINR)
    uint8_t *st_flags;      // 0 = dead candidate
    uint8_t *st_coll;
    int32_t *head;          // [n_rx]
    // SINR ticks whose frames are named by source indices (one start, one air time: all of a tick's frames overlap each other) sum
    // the interference per RECEIVER instead of keeping per-receiver lists: every candidate at the interference floor adds its
    // linear power in Q80 to acc[pos] (two words; exact whatever the order), a heard link's interference is that sum less its own
    // power, and bit 63 of acc_hi[pos] says that the receiver is on the air itself (half duplex).  nullptr: the lists.
    unsigned long long *acc_lo, *acc_hi; // [n_rx], zeroed by the tick's pre-pass
    // ordered records: A = (packet, engine position) order, B = (packet, node index) order
    uint32_t *out_count;    // [0] heard links stored, [1] dropped flag, [2] heard links total, [3] max per frame
    int32_t *a_pkt, *a_dst;
    uint8_t *a_verdict;
    double *a_rssi, *a_sinr, *a_prob;
    int32_t *out_pkt, *out_dst;
    uint8_t *out_verdict;
    double *out_rssi, *out_sinr, *out_prob;
    uint8_t *pkt_interference; // [n_new]
    // stochastic part
    uint32_t *draw_scan;    // [cap+1]
    uint32_t *scan_block;   // scratch for the generic scan
    uint64_t *rng_state;    // [1] java.util.Random state (48 bit)
    uint64_t *pkt_rng;      // [n_new] state before this rank's receiver draws of the packet
    uint32_t *pkt_draw_cnt; // [n_new] receiver draws of this rank for the packet
};

// rm_transmit's result block in host-mapped (pinned) memory, written by k_pack_result
constexpr int kTransmitMax = 2048;
struct TransmitResult {
    uint32_t stored, dropped, total, interference;
    uint32_t seq;                 // k_transmit_one: the call's sequence number, written after everything else
    uint32_t pad[3];
    int32_t dst[kTransmitMax];
    double rssi[kTransmitMax];
    double sinr[kTransmitMax];
    uint8_t verdict[kTransmitMax];
};

// A tick's result in host-mapped memory (rm_tick_flush*): header, then the arrays at the offsets the
// host computed for `links` records and `packets` packets.
struct HostHeader {
    uint32_t stored;      // records written (min(total, links))
    uint32_t dropped;     // the context's link capacity was exceeded
    uint32_t total;       // heard links of the tick
    uint32_t span_flag;   // a SINR tick held a frame outside its span
    uint32_t n_packets;
    uint32_t seq;         // written last
    uint32_t pad[10];
};
struct HostView {
    HostHeader *hdr;
    uint32_t *pkt_offset;       // [packets + 1]
    uint8_t *pkt_interference;  // [packets]
    int32_t *dst;               // (no packet column: a link's packet follows from pkt_offset)
    double *rssi, *sinr;        // rssi == nullptr: the medium hands the packet's transmit power through (UDGMRadioMedium.java:95,
                                // NullRadioMedium.java:57, N2NRadioMedium.java:51, UDGMConstantLossRadioMedium.java:22) -- it is sent once
                                // per PACKET (pkt_rssi), not with every link: 8 of a link's 13 bytes that need not cross PCIe
    double *pkt_rssi;           // [packets] the packets' transmit power
    uint8_t *verdict;
    uint32_t links, packets;    // room
};

// rm_batch_result_view: the results of several slots packed back to back into the host-mapped block.
struct PackSlot {
    TickDev t;
    int n_new, have_offsets;
    uint32_t pkt_base;      // first packet of this slot in the block's packet arrays (offsets: pkt_base + slot index)
    uint32_t pad;
};
struct BatchCounts {        // per slot, in the host-mapped block
    uint32_t stored, dropped, total, span_flag, link_base, pad[3];
};

// ---- reception stage (rm_events.hip): Simulator.generate*Events + processAllEvents + Transciever state on the device
// One transmitted packet whose events are still (partly) queued.  The pending packets form a window
// [pk_head, pk_tail) of a ring in transmission order; a packet's heard links sit contiguously in the link pool ring.
struct alignas(64) EvPacket {
    int64_t t0, t1;      // event times: max(start, currentTime at transmit), + air time (Simulator.java:323-333)
    int64_t gseq;        // packet number since rm_events_enable: its events' insertion order
    uint32_t link_off;   // first link in the pool (monotone counter, index = value & pool_mask)
    uint32_t link_cnt;
    int32_t src;
    int32_t lad0, lad1;  // ladder of its start / end events (rm_evorder.hpp)
    uint32_t flags;      // kEvStartDone | kEvDone | kEvImmediate | kEvNoTx
    uint32_t n_deliver;  // heard links with a delivery-mode end flank
    uint32_t pad[3];
};
constexpr uint32_t kEvStartDone = 1, kEvDone = 2, kEvImmediate = 4, kEvNoTx = 8;
enum { kEvRxStart = 0, kEvRxEnd = 1, kEvTxStart = 2, kEvTxEnd = 3 };

// The ring tails move with every appended tick.  They are kept twice: an append reads the set `par` (a kernel
// argument) and ONE thread writes the other one at its end -- no workgroup of the launch reads what another
// writes, so the launch needs no hand-off; the next launch is told the other parity.
struct EvTails {
    int64_t gseq_next;
    uint32_t pk_tail, pool_tail; // monotone counters
    uint32_t err;                // until the next drain has reported it: 1 packet ring full, 2 link pool full, 8 a tick was dropped for capacity
    uint32_t pad;
};
struct EvState {
    int64_t top_start, top_max; // rm_evorder.hpp (top_max == INT64_MIN: the top list is empty)
    int64_t t_prev;             // time of the last drain
    int32_t ladders, pad0;
    uint32_t pk_head, pool_head; // monotone counters
    uint32_t n_groups;          // fired (packet, phase) groups of the running drain
    uint32_t n_deliv;
    uint32_t err;               // until the drain has reported it: 4 group list full
    uint32_t done_a;            // "last workgroup" counter (node-info)
    uint32_t done_apply;        // ... of k_ev_apply (the last one finishes the drain)
    uint32_t n_dgroups;         // fired end groups that deliver something: the runs of the drain's delivery list
    EvTails tails[2];
    // (its own 128-byte line: k_ev_select's workgroups add to n_groups and take minima here at the same time)
    alignas(128) uint32_t first_live; // window index of the oldest packet with events still queued (0xFFFFFFFF: none)
    // k_ev_apply's "last workgroup" count in two levels: 512 workgroups that finish together on ONE word were six microseconds of
    // atomics in a row (a word takes ~88 per microsecond); sixteen words, a line each, then one (kEvDoneSub)
    alignas(128) uint32_t done_sub[16 * 32];
};
constexpr int kEvDoneSub = 16;

struct EvDev {
    EvState *st;
    EvPacket *pk;
    uint32_t pk_mask;
    int32_t *l_dst;
    double *l_rssi;
    uint8_t *l_verdict;
    uint32_t pool_mask;
    int64_t *g_time;      // fired groups: sort key (time, meta) ...
    uint64_t *g_meta;
    uint32_t *g_ref;      // ... packet ring index << 1 | phase (0 end, 1 start)
    uint32_t *g_rank, *cnt_by_rank, *off_by_rank;
    uint32_t *g_run;      // which run of the delivery list the (delivering end) group is
    unsigned long long *run_rec; // [2 * g_cap] per run: packet number, first delivery | count << 32 (copied to the host in one go at the end)
    uint32_t g_cap;
    unsigned long long *recv_key, *send_key; // [n] last writer of the node's receivingPacket / sendingPacket in the running drain
    uint8_t *receiving, *sending;            // [n] Transciever.receivingPacket / sendingPacket != null
    double *latched;                         // [n] Transciever.receivingRSSI
    int n_nodes;
    int own_first, own_count;                // nodes whose Tx / Rx events this context keeps (receiver partition by index range)
    const uint8_t *member;                   // spatial partition: [n_nodes] 1 = the node is a receiver of this context (else nullptr)
    int par;                                 // which EvState::tails are current
};

// a tick's heard links, packet by packet: the frames' segments (rm_tick.hip) or the compact arrays
struct EvLinkSrc {
    const int32_t *dst;
    const double *rssi;
    const uint8_t *verdict;
    const uint32_t *off;   // [n_new] first link of packet q (compact arrays: [n_new + 1], cnt == nullptr)
    const uint32_t *cnt;   // [n_new] or nullptr
    int n_scan;            // segments: entries of cnt to scan for the pool positions (n_cnt), 0 = off is the scan
    int per_frame_verdict; // every link of a frame carries the same verdict (no draws, no SINR: the frame's transmission failed or not)
};

// the delivery list of one drain in host-mapped memory
// Four quarters of 16 bytes, each written by ONE 16-byte store and each carrying the drain's sequence number: a quarter whose
// number is the expected one is complete (a store of 16 aligned bytes arrives as a whole), so the device need not wait for the
// fields to be acknowledged before it writes "the" number -- the host waits for all four (rm_events_process).
struct EvHeader {
    uint32_t seq;        // quarter 0
    uint32_t count;      // deliveries written
    uint32_t total;      // deliveries of the drain (count < total: the block was too small)
    uint32_t err;
    uint32_t seq1;       // quarter 1
    uint32_t pending_packets;
    uint32_t runs;       // runs of the delivery list (one per packet that delivers in this drain)
    uint32_t pad1;
    uint32_t seq2, pad2; // quarter 2
    int64_t next_packet; // number the next transmitted packet gets
    uint32_t seq3, pad3; // quarter 3
    int64_t oldest_packet; // number of the oldest packet with events still queued (== next_packet: none)
};
static_assert(sizeof(EvHeader) == 64, "one line of host-mapped memory");
// (the deliveries of one fired end group -- one packet -- are adjacent in the list: the packet number is sent once per run,
// not with each of its deliveries: 8 of 20 bytes that need not cross PCIe)
struct EvOut {
    EvHeader *hdr;
    int64_t *run_packet;  // [run_cap]
    uint32_t *run_first;  // [run_cap] first delivery of the run
    uint32_t *run_count;  // [run_cap]
    int32_t *dst;         // [cap]
    double *rssi;         // [cap]
    uint32_t cap, run_cap;
};
struct NodeInfoOut {
    uint32_t *seq;
    double *rssi;
    int32_t *receiving, *channel;
};
struct NodeChangeOut { // the nodes whose node-info changed since it was last reported (host-mapped)
    uint32_t *seq, *count;
    int32_t *node;
    double *rssi;
    int32_t *receiving, *channel;
};

struct LaunchCfg {
    bool f64_filter;  // fp32 frame too coarse: filter in fp64, no bounding boxes
    bool stochastic;  // java.util.Random draws may be consumed
    bool sorted;      // receiver table is spatially sorted (needs the per-packet reorder pass)
    bool bbox;        // use the bounding-box variant of the filter kernel
    bool shadow;      // second-level shadowing filter (table lookup on the link hash)
};

// kernels' host launchers (rm_filter.hip, rm_exact.hip, rm_reorder.hip, rm_transmit.hip)
hipError_t launch_prep_rx(hipStream_t s, const NodesDev &nd, const ModelDev &m);
hipError_t launch_patch_nodes(hipStream_t s, const NodesDev &nd, const NodePatch *dev_list, int n, const NodePatch &one);
hipError_t launch_pack_tx(hipStream_t s, const NodesDev &nd, const int32_t *dev_src, int n, int64_t start_us,
                          int64_t air_us, rm_tx_record *out);
hipError_t launch_pack_tx_batch(hipStream_t s, const NodesDev &nd, const int32_t *dev_src, int n_ticks, int n,
                                const int64_t *start_us, int64_t air_us, rm_tx_record *out, int world = 1); // dev_src / out: [world][n_ticks][n]
hipError_t launch_store_record(hipStream_t s, const rm_tx_record &r, rm_tx_record *dst);
hipError_t launch_pack_tick(hipStream_t s, const TickDev &t, int n_new, int pkt_shift_valid, const HostView &v, uint32_t *done_counter,
                            uint32_t seq);
hipError_t launch_fetch_ticks(hipStream_t s, const TickDev *host_mapped, int n, TickDev *dev_ticks);
hipError_t launch_pack_batch(hipStream_t s, const PackSlot *dev_slots, int n_slots, const HostView &v, BatchCounts *host_counts,
                             uint32_t *done_counter, uint32_t seq);
hipError_t launch_pack_result(hipStream_t s, const TickDev &t, TransmitResult *host_mapped);
hipError_t launch_transmit_one(hipStream_t s, const NodesDev &nd, const ModelDev &m, const rm_tx_record &tx,
                               uint64_t *rng_state, TransmitResult *host_mapped, uint32_t seq);
constexpr uint32_t kTransmitFallback = 0xFFFFFFFFu; // TransmitResult::total when k_transmit_one declined
hipError_t launch_filter(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                         const LaunchCfg &cfg);
// developer knobs read from the environment, once per API call that plans ticks (a batch plans up to 128 of them:
// three getenv per tick were most of its host time): RM_FILTER=grid|wg, RM_WG_RPT=1|2|4, RM_NO_SHADOW_TABLE
struct PlanKnobs {
    int filter;          // 0 = automatic, kFilterGrid + 1 / kFilterWg + 1 = forced
    int wg_rpt;          // 0 = automatic
    bool no_shadow_table;
    int batch_ticks;     // ticks of the batch being planned (0: a lone tick): a batch brings its own parallelism
};
PlanKnobs read_plan_knobs();
int plan_filter(TickDev &t, const LaunchCfg &cfg, bool want_wg, const PlanKnobs &knobs);
bool batch_eligible(const TickDev &t, const LaunchCfg &cfg, const ModelDev &m);
hipError_t launch_store_ticks(hipStream_t s, const TickDev *ticks, int n, TickDev *dev_ticks);
hipError_t launch_filter_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n,
                               const TickDev *dev_ticks, const LaunchCfg &cfg);
// (rm_filter.hip) the rank-level frame lists of a batch of gathered ticks: which of all ranks' frames can matter to THIS partition's
// receivers (m: the medium whose candidate level bounds the reach -- the SINR medium's interference floor where it has one);
// margin: metres added around the partition's boxes (frames that stay on the air: a receiver may move while they do);
// digests: the ranks' node-table digests as the all-gather left them (gather_base + r * gather_block + digest_off, two words
// each; digest_off < 0: none) -- a rank whose table differs from `mine` makes every tick of the batch RM_ERR_STATE
// Frames that stay on the air after they were selected for a partition (a batch of overlapping SINR ticks): the selection holds
// while every receiver of the partition stays inside the box it was made against -- the partition's box plus the margin.  Each
// such batch leaves that box and the time its last frame ends in a ring on the device; every later batch (and every lone tick
// over such a window) compares the partition's box AS IT IS NOW with the entries whose frames are still on the air, and a
// receiver that has left one of them -- it moved, its radio was switched on -- makes the tick RM_ERR_STATE instead of
// silently missing an interferer that was never kept for this rank.
struct CullEntry {
    float lo[3], hi[3];
    int64_t end_us;
};
constexpr int kCullRing = 16;
struct RankFramesArgs {
    const int32_t *gather_base;
    int world, gather_block, digest_off;
    uint64_t mine;
    float margin;
    int use_chmask;
    double sweep_level; // the candidate level of the sweep that follows (its pre-filter records are written here)
    CullEntry *ring;    // frames that stay on the air: the ring of boxes (nullptr: nothing of these ticks outlives them)
    int ring_slot;      // ... this batch's entry
    int64_t t_first, batch_end; // the batch's first t_begin, the end of its last frame
};
hipError_t launch_cull_check(hipStream_t s, const NodesDev &nd, const CullEntry *ring, int64_t t_begin, uint32_t *flag_word);
hipError_t launch_rank_frames(hipStream_t s, const NodesDev &nd, const ModelDev &m, TickDev *dev_ticks, int n, int max_frames, const RankFramesArgs &a);
constexpr int kGatherTrailer = RM_GATHER_TRAILER; // words behind a rank's source indices in its block of a sharded batch
hipError_t launch_stage_block(hipStream_t s, const int32_t *src, int n, uint64_t digest, int32_t *dst);
int filter_ticks_per_wg(const TickDev &t0, int n);
hipError_t launch_exact_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n,
                              const TickDev *dev_ticks, const LaunchCfg &cfg);
hipError_t launch_sinr_acc_batch(hipStream_t s, const ModelDev &m, int n, const TickDev *dev_ticks, int max_links, int share);
hipError_t launch_sinr_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n,
                             const TickDev *dev_ticks);
hipError_t launch_reorder_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n,
                                const TickDev *dev_ticks, const LaunchCfg &cfg);
hipError_t launch_batch_stage(hipStream_t s, int stage, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n,
                              const TickDev *dev_ticks, const LaunchCfg &cfg);
hipError_t launch_exact(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                        const LaunchCfg &cfg);
// the closed-loop tick (rm_tick.hip): one frame per workgroup, filter + exact evaluation in one launch
int frame_tick_segment(const TickDev &t, const LaunchCfg &cfg, const ModelDev &m);
bool frames_cand_applies(const TickDev &t, const LaunchCfg &cfg);
hipError_t launch_frames_cand(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg);
hipError_t launch_tick_frames(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg,
                              int seg_len, const ScanDev *scan = nullptr);
// (rm_airscan.hip) second launch of the SINR medium's lone tick by scan: interference sums, sinr and verdicts of the new frames' heard links
hipError_t launch_sinr_scan(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const ScanDev &sd, const LaunchCfg &cfg);
hipError_t launch_air_expire(hipStream_t s, rm_tx_record *recs, int n, int64_t t_seen); // (the on-air window when the clock goes back)
hipError_t launch_tick_frames_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n,
                                    const TickDev *dev_ticks, const LaunchCfg &cfg, int seg_len);
hipError_t launch_pack_frames(hipStream_t s, const ModelDev &m, const TickDev &t, int n_new, const HostView &v, uint32_t *done_counter,
                              uint32_t seq);
hipError_t launch_self_entries(hipStream_t s, const NodesDev &nd, const TickDev &t);
hipError_t launch_air_begin(hipStream_t s, const TickDev &t);
hipError_t launch_offsets(hipStream_t s, const TickDev &t);
hipError_t launch_sinr(hipStream_t s, const ModelDev &m, const TickDev &t);
hipError_t launch_finalize(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                           const LaunchCfg &cfg);
hipError_t launch_seg_scan(hipStream_t s, const TickDev &t);
hipError_t launch_reorder(hipStream_t s, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg);
hipError_t launch_draws_scan(hipStream_t s, const TickDev &t);
hipError_t launch_draw_nodes(hipStream_t s, const TickDev &t, int32_t *dev_nodes);
hipError_t launch_draws_apply_nodes(hipStream_t s, const ModelDev &m, const TickDev &t, const uint32_t *all_cnt, uint32_t *all_off,
                                    const int32_t *all_nodes, uint32_t stride, int world);
hipError_t launch_draws_batch(hipStream_t s, const ModelDev &m, const TickDev *ticks, int n, const TickDev *dev_ticks);
hipError_t launch_draws_apply(hipStream_t s, const ModelDev &m, const TickDev &t, const uint32_t *all_cnt, int world,
                              int rank);

// (rm_airbatch.hip) a batch of SINR ticks with frames that outlive their tick: index of the frames, then pairs / exact / verdicts
hipError_t launch_ov_begin(hipStream_t s, const OvTick *h_ticks, const int32_t *h_first, int n_ticks, int n_slots, OvTick *ticks,
                           int32_t *slot_first, uint32_t *misc, uint32_t *pair_tail);
hipError_t launch_ov_index(hipStream_t s, const NodesDev &nd, const ModelDev &m, const OvDev &ov, int max_slot_frames);
hipError_t launch_ov_sinr(hipStream_t s, const NodesDev &nd, const ModelDev &m, const OvDev &ov, int max_new, int max_links, const LaunchCfg &cfg);

// (rm_dense.hip) the tick of a medium in which a frame is heard by a large share of all nodes: node-order evaluation, ordered compaction
bool dense_tick_applies(const TickDev &t, const LaunchCfg &cfg, const ModelDev &m, const NodesDev &nd, bool whole_or_range);
int dense_tick_cells(const NodesDev &nd, const TickDev &t);
hipError_t launch_dense_tick(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, uint32_t *cell_cnt, uint32_t *cell_off,
                             unsigned long long *cell_mask, bool lazy_write); // cell_mask: 16 lane masks per (frame, chunk) cell
hipError_t launch_dense_layout(hipStream_t s, const ModelDev &m, const TickDev &t, const uint32_t *cell_cnt, uint32_t *cell_off, int chunks);
hipError_t launch_dense_write(hipStream_t s, const ModelDev &m, const TickDev &t, const uint32_t *cell_cnt, const uint32_t *cell_off,
                              const unsigned long long *cell_mask, int rx_first, int chunks);

// reception stage (rm_events.hip)
hipError_t launch_ev_append(hipStream_t s, const EvDev &e, const EvLinkSrc &ls, const rm_tx_record *tx, int n_new, int64_t now,
                            int immediate, const uint32_t *dropped_flag);
// window: the host's bound on the pending packets (0: unknown)
// (fresh_*: the tick whose append has been left for this drain -- append and selection in one launch; e.par is then the parity
// that append reads)
hipError_t launch_ev_drain(hipStream_t s, const EvDev &e, const EvOut &out, int64_t time_us, uint32_t seq, uint32_t window,
                           const EvLinkSrc *fresh_ls = nullptr, const rm_tx_record *fresh_tx = nullptr, int fresh_n = 0, int64_t fresh_now = 0,
                           int fresh_immediate = 0, const uint32_t *fresh_dropped = nullptr);
hipError_t launch_node_info(hipStream_t s, const EvDev &e, const NodesDev &nd, const int32_t *dev_nodes, int n, double base_rssi,
                            const NodeInfoOut &out, uint32_t seq);
hipError_t launch_node_info_changed(hipStream_t s, const EvDev &e, const NodesDev &nd, int n, double base_rssi, double *rep_rssi,
                                    int2 *rep_sc, const NodeChangeOut &out, uint32_t cap, uint32_t seq, uint32_t *counters);

// host-side mirrors of device math used for constants (rm_math.hpp, exported by rm_transmit.hip)
double host_det_pow10(double y);
double host_det_math(int fn, double x);                                  // rm_det_math (test hook)
uint64_t host_link_hash(uint64_t seed, uint32_t a, uint32_t b, double *u); // rm_link_hash (test hook)
uint64_t host_mix64(uint64_t z);
void host_lcg_jump_map(uint64_t steps, uint64_t *A, uint64_t *C);

} // namespace rm
