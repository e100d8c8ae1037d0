// rm_dense.hip -- the tick of a medium in which a frame is heard by a large share of all nodes: the reference's DEFAULT
// medium (NullRadioMedium.java:47-77: every same-channel node hears everything), an N2N matrix without loss, a unit-disc
// range that covers most of the field (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math)
//
// Everything the other paths are built around -- a conservative cull, a candidate list, ranking a frame's few dozen heard
// links by node index -- has nothing to do here: there is nothing to cull, and the output IS the node table, frame by
// frame.  So a frame's receivers are visited in NODE-INDEX order, the reference's own visiting order
// (Simulator.getNodes(), UDGMRadioMedium.java:95): one exact evaluation per (frame, node) from the node-ordered columns of
// the source table (coalesced: 5 bytes per node for the Null medium, 37 for the unit disc) and an ordered compaction; what
// is left is record writing -- 17 bytes per heard link (25 with the SINR column) in runs of whole cache lines.
//   k_dense_count   (chunk of 1024 consecutive nodes, eight frames): the chunk's columns once, in registers; per frame the heard
//                   links as 16 lane masks (a wave's 64 consecutive nodes each) and their number
//                   -- where the tick ENDS by default: masks and counts are its result (rm_result_dense); what is derived from them is
//                   derived when somebody asks (launch_dense_layout: offsets and totals; launch_dense_write: the records)
//   k_dense_write   (frame, chunk): the records of the masks' set bits.  A lane's node is 64 * k + lane of the wave's 256, so
//                   every store instruction of a wave writes consecutive records: whole lines, one array at a time.  The
//                   chunk's offset is the sum of the counts before it, taken by the workgroup itself from the (L2-resident)
//                   counts -- up to 8192 (frame, chunk) cells; beyond that k_dense_scan (one workgroup) lays them out first.
// History.  The fixed-segment tick of rm_tick.hip took 0.31 ms for 200 frames x 20 k nodes (one workgroup per frame, two
// evaluations and a bitmap pass over the node indices in LDS).  Round 4's first form of this file -- count, one-workgroup scan,
// write with the evaluation repeated, four consecutive nodes per thread -- took 14 + 10.5 + 35.5 us for the Null medium on that
// shape (28 + 10.5 + 47 for the unit disc).  A one-launch form -- offsets by decoupled look-back over (frame, chunk) states, in
// one and in two levels -- was built and measured at 75-84 us; it is not in the tree.
#include "rm_device.hpp"

#include <stdlib.h>

namespace rm {

constexpr int kDnPer = 4;                  // nodes per thread: 64 * k + lane of the wave's 256 consecutive nodes
constexpr int kDnChunk = 256 * kDnPer;     // nodes per workgroup
constexpr int kDnFrames = 8;               // frames per workgroup of the count pass
constexpr int kDnFusedCells = 8192;        // up to here the write pass sums the counts before its cell itself

// a node's columns as the media read them (what a medium does not read is not loaded)
struct DnNode {
    double x, y, z, rxprob;
    int channel, int_id;
    bool enabled;
};

template <int MODEL>
RM_D DnNode dense_node(const NodesDev &nd, int j, bool valid)
{
    DnNode n{};
    n.enabled = false;
    if (!valid) return n;
    n.enabled = nd.senabled[j] != 0;
    n.channel = nd.schannel[j];
    if (MODEL == RM_MODEL_N2N) {
        n.int_id = nd.sint_id[j];
        n.rxprob = nd.srxprob[j];
    }
    if (MODEL == RM_MODEL_UDGM || MODEL == RM_MODEL_UDGM_CONST) {
        n.x = nd.sx[j], n.y = nd.sy[j], n.z = nd.sz[j];
        if (MODEL == RM_MODEL_UDGM) n.rxprob = nd.srxprob[j];
    }
    return n;
}

// The reference's tests for one (frame, node) link, in its order, from the node-ordered columns (eval_link's arithmetic:
// the same helpers, one rounding at a time).  The dense tick runs only where no draw can happen, so a heard link is
// delivered unless the frame's transmission failed (txSuccess <= 0), and its rssi is the packet's transmit power (the
// reference media hand it through).
template <int MODEL>
RM_D bool dense_eval(const ModelDev &m, const rm_tx_record &tx, int src_int_id, int j, const DnNode &n)
{
    if (tx.src < 0 || j == tx.src) return false;       // padding record; node != source
    if (!n.enabled) return false;                       // radio.isEnabled()
    if (n.channel != tx.channel) return false;          // radio.getWirelessChannel() == channel
    if (MODEL == RM_MODEL_NULL) return true;
    if (MODEL == RM_MODEL_N2N) { // N2NRadioMedium.java:28-37
        const int sid = src_int_id, did = n.int_id;
        double p = 0.0;
        if (m.n2n != nullptr && sid > 0 && did > 0 && sid <= m.n2n_m && did <= m.n2n_m) p = m.n2n[int64_t(sid - 1) * m.n2n_m + (did - 1)] * n.rxprob;
        return p > 0.0;
    }
    // Far from the range's edge the outcome needs no square root and no division (most of this kernel's issue time, at
    // 4 M links per tick): with s = the sum of squares ref_distance takes the root of, the reference's d * d is s to
    // 4 ulp, so outside a relative band of 1e-9 around range^2 the comparison is decided by s itself; and inside the
    // range, ratio <= 1 - 1e-9 leaves 1 - ratio * (1 - ratio_rx) >= 9e-10 for ratio_rx in [0, 1], so the probability is
    // positive exactly when the receiver's is (kept away from the subnormals).  Everything else takes the reference's steps.
    {
        double dx = tx.x - n.x, dy = tx.y - n.y, dz = tx.z - n.z;
        dx = dx * dx, dy = dy * dy, dz = dz * dz;
        const double s = dx + dy + dz;
        const double range = (MODEL == RM_MODEL_UDGM_CONST) ? m.const_range : m.udgm_range;
        const double r2 = range * range;
        if (range > 0.0) {
            if (s > r2 * (1.0 + 1e-9)) return false;
            if (s < r2 * (1.0 - 1e-9)) {
                if (MODEL == RM_MODEL_UDGM_CONST) return true;
                if (m.udgm_ratio_rx >= 0.0 && m.udgm_ratio_rx <= 1.0) {
                    if (n.rxprob >= 1e-280) return true;
                    if (n.rxprob <= 0.0) return false;
                }
            }
        }
    }
    const double d = ref_distance(tx.x, tx.y, tx.z, n.x, n.y, n.z);
    if (MODEL == RM_MODEL_UDGM_CONST) return d < m.const_range;
    // UDGMRadioMedium.java:67-81 ; Math.pow(v, 2.0) == v*v
    const double d2 = d * d, dmax = m.udgm_range;
    if (dmax == 0.0) return false;
    double ratio = d2 / (dmax * dmax);
    if (ratio > 1.0) return false;
    ratio = 1.0 - ratio * (1.0 - m.udgm_ratio_rx);
    return ratio * n.rxprob > 0.0;
}

// the frame's record: given, or built from its source index (the count pass leaves it where every later reader looks)
RM_D rm_tx_record dense_frame(const NodesDev &nd, const TickDev &t, int q)
{
    return t.src_list ? make_tx_record(nd, t.src_list[q], t.src_start_us, t.src_air_us) : t.tx[t.first_new + q];
}

// what every tick leaves for the one that follows and for its readers, whichever kernel lays the cells out
RM_D void dense_tick_tail(const ModelDev &m, const TickDev &t, int tid, int n_threads, bool with_interference = true)
{
    for (int i = tid; i < t.shift; i += n_threads) t.slot_off[i] = 0u;
    if (with_interference) write_pkt_interference(m, t, tid, n_threads);
    // what the sweep's first kernel does for the tick that follows (the other parity's counters start at zero)
    if (tid < 8) t.next_counters[tid] = 0u;
    for (int i = tid; i < kShards; i += n_threads) t.next_shard_count[i * kShardStride] = 0u;
    if (!t.use_matrix)
        for (int i = tid; i < t.zero_len; i += n_threads) {
            t.cursor[i] = 0u;
            t.cand_tot_next[i] = 0u;
        }
}
template <int MODEL>
__global__ void __launch_bounds__(256)
k_dense_count(const NodesDev nd, const ModelDev m, const TickDev t, uint32_t *cell_cnt, unsigned long long *cell_mask, int chunks, const int with_tail)
{
    __shared__ uint32_t s_w[kDnFrames][4];
    const int chunk = blockIdx.x, q0 = int(blockIdx.y) * kDnFrames;
    const int n_new = t.n_active - t.first_new;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jw = nd.rx_first + chunk * kDnChunk + wave * (64 * kDnPer);
    const int j_end = nd.rx_first + nd.pos_span;
    // the workgroup's frames first, one per thread: a record named by its source index is two dependent round trips, and eight
    // of them one after the other were most of this kernel's time
    __shared__ rm_tx_record s_tx[kDnFrames];
    __shared__ int s_sid[kDnFrames];
    if (threadIdx.x < kDnFrames && q0 + int(threadIdx.x) < n_new) {
        const int q = q0 + int(threadIdx.x);
        const rm_tx_record tx = dense_frame(nd, t, q);
        if (t.src_list && chunk == 0) t.tx_build[t.first_new + q] = tx;
        // (a caller's record with a transmit probability the node table does not have, in a tick planned without draws:
        // reported when the result is read, as the other forms of the tick do)
        if (chunk == 0 && t.check_txprob && tx.src >= 0 && tx.txprob > 0.0 && tx.txprob < 1.0) t.stage_count[6] = 2u;
        s_tx[threadIdx.x] = tx;
        s_sid[threadIdx.x] = (MODEL == RM_MODEL_N2N && tx.src >= 0) ? nd.sint_id[tx.src] : 0;
        if (with_tail && chunk == 0) { // the packet's Tx-failure flag, from the record in hand (write_pkt_interference's expression)
            const bool draws_possible = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_N2N || m.kind == RM_MODEL_LOGDIST);
            t.pkt_interference[q] = (draws_possible && tx_success(m, tx) <= 0.0) ? 1 : 0;
        }
    }
    DnNode node[kDnPer];
#pragma unroll
    for (int k = 0; k < kDnPer; ++k) node[k] = dense_node<MODEL>(nd, jw + k * 64 + lane, jw + k * 64 + lane < j_end);
    __syncthreads();
    for (int f = 0; f < kDnFrames; ++f) { // block-uniform
        const int q = q0 + f;
        uint32_t cnt = 0;
        if (q < n_new) {
            const rm_tx_record tx = s_tx[f];
            const int sid = s_sid[f];
            const size_t cell = size_t(q) * size_t(chunks) + size_t(chunk);
#pragma unroll
            for (int k = 0; k < kDnPer; ++k) {
                const uint64_t hm = ballot64(dense_eval<MODEL>(m, tx, sid, jw + k * 64 + lane, node[k]));
                if (lane == 0) cell_mask[cell * 16u + uint32_t(wave * kDnPer + k)] = hm;
                cnt += uint32_t(__popcll(hm));
            }
        }
        if (lane == 0) s_w[f][wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < kDnFrames && q0 + int(threadIdx.x) < n_new) {
        const int f = threadIdx.x;
        cell_cnt[size_t(q0 + f) * size_t(chunks) + size_t(chunk)] = s_w[f][0] + s_w[f][1] + s_w[f][2] + s_w[f][3];
    }
    // a tick that ends with its cells (the layout and the records on request): what the tick that follows relies on is left here
    if (with_tail && blockIdx.x == 0 && blockIdx.y == 0) dense_tick_tail(m, t, threadIdx.x, 256, false); // (the flags: by the frames' own workgroups, above)
}

RM_D void dense_tick_total(const TickDev &t, uint32_t total, int tid, int n_threads)
{
    const int n_new = t.n_active - t.first_new;
    for (int i = t.shift + n_new + tid; i <= t.n_cnt; i += n_threads) t.slot_off[i] = total; // (the padding slots are empty)
    if (tid == 0) {
        t.out_count[0] = min(total, t.cap);
        t.out_count[1] = (total > t.cap || t.stage_count[1] != 0u) ? 1u : 0u;
        t.out_count[2] = total;
        t.out_count[3] = 0u;
    }
}

// one workgroup (ticks of more than kDnFusedCells cells): exclusive scan of the cells (frame-major: a frame's chunks in node
// order), packet offsets, counters
__global__ void __launch_bounds__(1024) k_dense_scan(const ModelDev m, const TickDev t, const uint32_t *cell_cnt, uint32_t *cell_off, int chunks, const int with_tail)
{
    __shared__ uint32_t s_wave[16];
    const int n_new = t.n_active - t.first_new;
    const int cells = n_new * chunks;
    if (cells <= 8 * 1024) {
        // up to eight consecutive cells per thread, all asked for at once: one round trip, one scan (the loop below is a chain of
        // a round trip and two barriers per 1024 cells -- 6.6 us for the 4000 cells of the bench's tick, as long as the count pass)
        const int i0 = int(threadIdx.x) * 8;
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = (i0 + k < cells) ? cell_cnt[i0 + k] : 0u;
            sum += v[k];
        }
        if (with_tail) dense_tick_tail(m, t, threadIdx.x, 1024); // (its loads and stores depend on nothing here: they fly under the scan)
        uint32_t total;
        uint32_t run = block_exclusive_scan_1024(sum, s_wave, total);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k;
            if (i < cells) {
                cell_off[i] = run;
                if (i % chunks == 0) t.slot_off[t.shift + i / chunks] = run; // a frame's first cell: its packet offset
            }
            run += v[k];
        }
        dense_tick_total(t, total, threadIdx.x, 1024);
        return;
    }
    uint32_t carry = 0;
    for (int base = 0; base < cells; base += 1024) {
        const int i = base + int(threadIdx.x);
        const uint32_t v = (i < cells) ? cell_cnt[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
        if (i < cells) {
            cell_off[i] = carry + ex;
            if (i % chunks == 0) t.slot_off[t.shift + i / chunks] = carry + ex; // a frame's first cell: its packet offset
        }
        carry += total;
    }
    dense_tick_total(t, carry, threadIdx.x, 1024);
    if (with_tail) dense_tick_tail(m, t, threadIdx.x, 1024);
}

template <bool FUSED>
__global__ void __launch_bounds__(256)
k_dense_write(const ModelDev m, const TickDev t, const uint32_t *cell_cnt, const uint32_t *cell_off, const unsigned long long *cell_mask,
              int rx_first, int chunks)
{
    __shared__ uint32_t s_w[4];
    const int q = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cell = uint32_t(q) * uint32_t(chunks) + uint32_t(chunk);
    // the cell's sixteen masks, one per lane (and again in the lanes above: every lane has company to shuffle with)
    const unsigned long long mk = cell_mask[size_t(cell) * 16u + uint32_t(lane & 15)];
    const rm_tx_record tx = t.tx[t.first_new + q]; // (built by the count pass when the tick named its frames by source index)
    uint32_t base;
    if (FUSED) { // the counts of the cells before this one
        uint32_t part = 0;
        for (uint32_t i = threadIdx.x; i < cell; i += 256u) part += cell_cnt[i];
        for (int d = 32; d >= 1; d >>= 1) part += uint32_t(__shfl_xor(int(part), d));
        if (lane == 0) s_w[wave] = part;
        __syncthreads();
        base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    } else {
        base = cell_off[cell];
    }
    const uint32_t pc = uint32_t(__popcll(mk));
    const uint32_t inc = wave_inclusive_scan((lane < 16) ? pc : 0u, lane); // (lanes 0..15: the cell's masks in node order)
    if (FUSED) {
        if (chunk == 0 && threadIdx.x == 0) t.slot_off[t.shift + q] = base; // a frame's first cell: its packet offset
        if (cell + 1u == uint32_t(t.n_active - t.first_new) * uint32_t(chunks)) { // the last cell knows the tick's total
            const uint32_t total = base + uint32_t(__shfl(int(inc), 15));
            dense_tick_total(t, total, threadIdx.x, 256);
        }
        if (cell == 0u) dense_tick_tail(m, t, threadIdx.x, 256);
    }
    const bool draws_possible = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_N2N || m.kind == RM_MODEL_LOGDIST);
    const uint8_t verdict = (draws_possible && tx_success(m, tx) <= 0.0) ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
    const int jw = rx_first + chunk * kDnChunk + wave * (64 * kDnPer);
#pragma unroll
    for (int k = 0; k < kDnPer; ++k) {
        const int i = wave * kDnPer + k;
        const unsigned long long hm = (unsigned long long)__shfl((long long)mk, i);
        const uint32_t before = uint32_t(__shfl(int(inc - ((lane < 16) ? pc : 0u)), i));
        if (!((hm >> lane) & 1ull)) continue;
        const uint32_t o = base + before + lane_prefix(hm);
        if (o < t.cap) { // consecutive lanes, consecutive records: every store instruction writes whole lines of one array
            t.out_pkt[o] = q;
            t.out_dst[o] = jw + k * 64 + lane;
            t.out_verdict[o] = verdict;
            t.out_rssi[o] = tx.txpower; // reference media hand the packet's transmit power through
            if (t.out_sinr) t.out_sinr[o] = 0.0;
        }
    }
}

// Is this tick one for the dense form?  No draws, no SINR, the whole node table or an index range of it as receivers, and
// a medium in which a frame reaches a large share of them: no geometry at all (Null, N2N), or a unit disc that covers
// at least a sixteenth of the table's bounding square and more nodes than a frame's fixed segment holds.
bool dense_tick_applies(const TickDev &t, const LaunchCfg &cfg, const ModelDev &m, const NodesDev &nd, bool whole_or_range)
{
    const char *e_knob = getenv("RM_DENSE_TICK"); // 0: never, 1: whenever the configuration allows it (read per tick: tests switch it)
    const int knob = e_knob ? atoi(e_knob) : -1;
    if (knob == 0) return false;
    if (cfg.stochastic || !whole_or_range || t.air_scan || t.air.pool != nullptr || t.first_eval != t.first_new) return false;
    if (m.kind == RM_MODEL_LOGDIST) return false; // (a frame's cut-off depends on its power; the culled paths keep it)
    if (t.n_active - t.first_new <= 0 || t.n_active - t.first_new > 65535 || t.n_rx <= 0 || nd.srxprob == nullptr) return false;
    if (long(t.n_active - t.first_new) * long(cdiv(nd.pos_span, kDnChunk)) > (1L << 24)) return false; // (cells of the scan)
    if (knob == 1) return true;
    if (m.kind == RM_MODEL_NULL || m.kind == RM_MODEL_N2N) return true;
    const double cut = m.geo_cut;
    if (!(cut > 0.0)) return false;
    const double side = 2.0 * m.coord_bound;
    const double share = (side > 0.0) ? std::min(1.0, 3.14159265358979 * cut * cut / (side * side)) : 1.0;
    return share >= 1.0 / 16.0 && share * double(t.n_rx) > double(kFrameSegMax);
}

// the record pass on its own (a tick whose records were left for whoever asks for them: launch_dense_tick with lazy_write)
hipError_t launch_dense_write(hipStream_t s, const ModelDev &m, const TickDev &t, const uint32_t *cell_cnt, const uint32_t *cell_off,
                              const unsigned long long *cell_mask, int rx_first, int chunks)
{
    const int n_new = t.n_active - t.first_new;
    if (n_new <= 0 || chunks <= 0) return hipSuccess;
    RM_KLAUNCH((k_dense_write<false>), dim3(chunks, n_new), dim3(256), 0, s, m, t, cell_cnt, cell_off, cell_mask, rx_first, chunks);
    return hipGetLastError();
}

// the cells' layout on its own (a tick that ended with its cells): cell and packet offsets, the totals
hipError_t launch_dense_layout(hipStream_t s, const ModelDev &m, const TickDev &t, const uint32_t *cell_cnt, uint32_t *cell_off, int chunks)
{
    if (t.n_active - t.first_new <= 0 || chunks <= 0) return hipSuccess;
    RM_KLAUNCH(k_dense_scan, dim3(1), dim3(1024), 0, s, m, t, cell_cnt, cell_off, chunks, 0);
    return hipGetLastError();
}

// lazy_write: the tick ends with its cells -- the heard links of every (frame, 1024 nodes) cell as sixteen lane masks and the
// cells' counts: ONE launch.  That IS the result of such a medium: a link's rssi is its packet's transmit power, its verdict its
// packet's.  What is derived from the cells is derived when somebody asks: the layout (cell and packet offsets, totals:
// launch_dense_layout -- rm_result_dense, rm_result_count) and the 17-byte records, 68 MB for 4 M links (materialize ->
// launch_dense_write).
hipError_t launch_dense_tick(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, uint32_t *cell_cnt, uint32_t *cell_off,
                             unsigned long long *cell_mask, bool lazy_write)
{
    const int n_new = t.n_active - t.first_new;
    const int chunks = cdiv(nd.pos_span, kDnChunk);
    if (n_new <= 0 || chunks <= 0) return hipSuccess;
    const dim3 grid_c(chunks, cdiv(n_new, kDnFrames)), grid(chunks, n_new), block(256);
    switch (m.kind) {
    case RM_MODEL_NULL: RM_KLAUNCH((k_dense_count<RM_MODEL_NULL>), grid_c, block, 0, s, nd, m, t, cell_cnt, cell_mask, chunks, lazy_write ? 1 : 0); break;
    case RM_MODEL_UDGM: RM_KLAUNCH((k_dense_count<RM_MODEL_UDGM>), grid_c, block, 0, s, nd, m, t, cell_cnt, cell_mask, chunks, lazy_write ? 1 : 0); break;
    case RM_MODEL_UDGM_CONST: RM_KLAUNCH((k_dense_count<RM_MODEL_UDGM_CONST>), grid_c, block, 0, s, nd, m, t, cell_cnt, cell_mask, chunks, lazy_write ? 1 : 0); break;
    case RM_MODEL_N2N: RM_KLAUNCH((k_dense_count<RM_MODEL_N2N>), grid_c, block, 0, s, nd, m, t, cell_cnt, cell_mask, chunks, lazy_write ? 1 : 0); break;
    default: return hipErrorInvalidValue;
    }
    // (the count pass's last workgroup laying the cells out itself -- one launch instead of two -- was built and measured: every
    // workgroup's release fence writes its XCD's L2 back, 47 us for the count pass instead of 5.6 + 6.6 for the two launches)
    if (lazy_write) {
        // (nothing more: the layout is one more launch and a tick's worth of latency that a reader of the masks never needs)
    } else if (long(n_new) * long(chunks) <= long(kDnFusedCells)) {
        RM_KLAUNCH((k_dense_write<true>), grid, block, 0, s, m, t, cell_cnt, cell_off, cell_mask, nd.rx_first, chunks);
    } else {
        RM_KLAUNCH(k_dense_scan, dim3(1), dim3(1024), 0, s, m, t, cell_cnt, cell_off, chunks, 1);
        RM_KLAUNCH((k_dense_write<false>), grid, block, 0, s, m, t, cell_cnt, cell_off, cell_mask, nd.rx_first, chunks);
    }
    return hipGetLastError();
}

int dense_tick_cells(const NodesDev &nd, const TickDev &t) { return (t.n_active - t.first_new) * cdiv(nd.pos_span, kDnChunk); }

} // namespace rm
