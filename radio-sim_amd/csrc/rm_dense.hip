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
//   k_dense_count   (frame, chunk of 1024 consecutive nodes): the links heard in the chunk
//   k_dense_scan    one workgroup: the chunks' offsets (frame-major), the packet offsets, the tick's counters
//   k_dense_write   the same evaluation again (cheaper than keeping four million verdicts between the launches), every
//                   thread's heard links to offset(cell) + rank in the chunk: consecutive threads write consecutive records
// The fixed-segment tick of rm_tick.hip took 0.31 ms for 200 frames x 20 k nodes (one workgroup per frame, two evaluations
// and a bitmap pass over the node indices in LDS).  A one-launch form of this file -- the chunks' offsets by decoupled
// look-back over (frame, chunk) states, in one and in two levels -- was built and measured at 75-84 us for the Null medium
// against 67 us for these three launches with the same evaluation; it is not in the tree.
#include "rm_device.hpp"

#include <stdlib.h>

namespace rm {

constexpr int kDnPer = 4;                  // consecutive nodes per thread
constexpr int kDnChunk = 256 * kDnPer;     // nodes per workgroup

// The reference's tests for one (frame, node) link, in its order, from the node-ordered columns (eval_link's arithmetic:
// the same helpers, one rounding at a time).  The dense tick runs only where no draw can happen, so a heard link is
// delivered unless the frame's transmission failed (txSuccess <= 0).
template <int MODEL>
RM_D bool dense_eval(const ModelDev &m, const NodesDev &nd, const rm_tx_record &tx, int j, double &rssi)
{
    if (tx.src < 0 || j == tx.src) return false;       // padding record; node != source
    if (!nd.senabled[j]) return false;                  // radio.isEnabled()
    if (nd.schannel[j] != tx.channel) return false;     // radio.getWirelessChannel() == channel
    rssi = tx.txpower;                                  // reference media hand the packet's transmit power through
    if (MODEL == RM_MODEL_NULL) return true;
    if (MODEL == RM_MODEL_N2N) { // N2NRadioMedium.java:28-37
        const int sid = nd.sint_id[tx.src], did = nd.sint_id[j];
        double p = 0.0;
        if (m.n2n != nullptr && sid > 0 && did > 0 && sid <= m.n2n_m && did <= m.n2n_m) p = m.n2n[int64_t(sid - 1) * m.n2n_m + (did - 1)] * nd.srxprob[j];
        return p > 0.0;
    }
    const double d = ref_distance(tx.x, tx.y, tx.z, nd.sx[j], nd.sy[j], nd.sz[j]);
    if (MODEL == RM_MODEL_UDGM_CONST) return d < m.const_range;
    // UDGMRadioMedium.java:67-81 ; Math.pow(v, 2.0) == v*v
    const double d2 = d * d, dmax = m.udgm_range;
    if (dmax == 0.0) return false;
    double ratio = d2 / (dmax * dmax);
    if (ratio > 1.0) return false;
    ratio = 1.0 - ratio * (1.0 - m.udgm_ratio_rx);
    return ratio * nd.srxprob[j] > 0.0;
}

// the frame's record: given, or built from its source index (the count pass leaves it where every later reader looks)
RM_D rm_tx_record dense_frame(const NodesDev &nd, const TickDev &t, int q)
{
    return t.src_list ? make_tx_record(nd, t.src_list[q], t.src_start_us, t.src_air_us) : t.tx[t.first_new + q];
}

template <int MODEL>
__global__ void __launch_bounds__(256) k_dense_count(const NodesDev nd, const ModelDev m, const TickDev t, uint32_t *cell_cnt, int chunks)
{
    __shared__ uint32_t s_w[4];
    const int q = blockIdx.y, chunk = blockIdx.x;
    const rm_tx_record tx = dense_frame(nd, t, q);
    if (t.src_list && chunk == 0 && threadIdx.x == 0) t.tx_build[t.first_new + q] = tx;
    const int j0 = nd.rx_first + chunk * kDnChunk + int(threadIdx.x) * kDnPer;
    const int j_end = nd.rx_first + nd.pos_span;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kDnPer; ++k) {
        double rssi;
        if (j0 + k < j_end && dense_eval<MODEL>(m, nd, tx, j0 + k, rssi)) ++cnt;
    }
    for (int d = 32; d >= 1; d >>= 1) cnt += uint32_t(__shfl_xor(int(cnt), d));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) cell_cnt[size_t(q) * size_t(chunks) + size_t(chunk)] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// one workgroup: exclusive scan of the cells (frame-major: a frame's chunks in node order), packet offsets, counters
__global__ void __launch_bounds__(1024) k_dense_scan(const ModelDev m, const TickDev t, const uint32_t *cell_cnt, uint32_t *cell_off, int chunks)
{
    __shared__ uint32_t s_wave[16];
    const int n_new = t.n_active - t.first_new;
    const int cells = n_new * chunks;
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
    for (int i = t.shift + n_new + int(threadIdx.x); i <= t.n_cnt; i += 1024) t.slot_off[i] = carry; // (the padding slots are empty)
    for (int i = int(threadIdx.x); i < t.shift; i += 1024) t.slot_off[i] = 0u;
    write_pkt_interference(m, t, threadIdx.x, 1024);
    // what the sweep's first kernel does for the tick that follows (the other parity's counters start at zero)
    if (threadIdx.x < 8) t.next_counters[threadIdx.x] = 0u;
    if (threadIdx.x < uint32_t(kShards)) t.next_shard_count[threadIdx.x * kShardStride] = 0u;
    if (!t.use_matrix)
        for (int i = int(threadIdx.x); i < t.zero_len; i += 1024) {
            t.cursor[i] = 0u;
            t.cand_tot_next[i] = 0u;
        }
    if (threadIdx.x == 0) {
        t.out_count[0] = min(carry, t.cap);
        t.out_count[1] = (carry > t.cap || t.stage_count[1] != 0u) ? 1u : 0u;
        t.out_count[2] = carry;
        t.out_count[3] = 0u;
    }
}

template <int MODEL>
__global__ void __launch_bounds__(256) k_dense_write(const NodesDev nd, const ModelDev m, const TickDev t, const uint32_t *cell_off, int chunks)
{
    __shared__ uint32_t s_w[4];
    const int q = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const rm_tx_record tx = t.tx[t.first_new + q]; // (built by the count pass when the tick named its frames by source index)
    const int j0 = nd.rx_first + chunk * kDnChunk + int(threadIdx.x) * kDnPer;
    const int j_end = nd.rx_first + nd.pos_span;
    bool heard[kDnPer];
    double rssi[kDnPer];
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kDnPer; ++k) {
        rssi[k] = 0.0;
        heard[k] = j0 + k < j_end && dense_eval<MODEL>(m, nd, tx, j0 + k, rssi[k]);
        cnt += heard[k] ? 1u : 0u;
    }
    const uint32_t inc = wave_inclusive_scan(cnt, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t o = cell_off[size_t(q) * size_t(chunks) + size_t(chunk)] + inc - cnt;
    for (int w = 0; w < wave; ++w) o += s_w[w];
    const bool draws_possible = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_N2N || m.kind == RM_MODEL_LOGDIST);
    const uint8_t verdict = (draws_possible && tx_success(m, tx) <= 0.0) ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
#pragma unroll
    for (int k = 0; k < kDnPer; ++k) {
        if (!heard[k]) continue;
        if (o < t.cap) {
            // (plain stores: the L2 puts a wave's 4-, 16- and 32-byte pieces together into whole lines; nontemporal stores of
            // the same records took 100 us instead of 31)
            t.out_pkt[o] = q;
            t.out_dst[o] = j0 + k;
            t.out_verdict[o] = verdict;
            t.out_rssi[o] = rssi[k];
            if (t.out_sinr) t.out_sinr[o] = 0.0;
        }
        ++o;
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

hipError_t launch_dense_tick(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, uint32_t *cell_cnt, uint32_t *cell_off)
{
    const int n_new = t.n_active - t.first_new;
    const int chunks = cdiv(nd.pos_span, kDnChunk);
    if (n_new <= 0 || chunks <= 0) return hipSuccess;
    const dim3 grid(chunks, n_new), block(256);
#define RM_DN(MODEL)                                                                                   \
    do {                                                                                               \
        RM_KLAUNCH((k_dense_count<MODEL>), grid, block, 0, s, nd, m, t, cell_cnt, chunks);             \
        RM_KLAUNCH(k_dense_scan, dim3(1), dim3(1024), 0, s, m, t, cell_cnt, cell_off, chunks);         \
        RM_KLAUNCH((k_dense_write<MODEL>), grid, block, 0, s, nd, m, t, cell_off, chunks);             \
    } while (0)
    switch (m.kind) {
    case RM_MODEL_NULL: RM_DN(RM_MODEL_NULL); break;
    case RM_MODEL_UDGM: RM_DN(RM_MODEL_UDGM); break;
    case RM_MODEL_UDGM_CONST: RM_DN(RM_MODEL_UDGM_CONST); break;
    case RM_MODEL_N2N: RM_DN(RM_MODEL_N2N); break;
    default: return hipErrorInvalidValue;
    }
#undef RM_DN
    return hipGetLastError();
}

int dense_tick_cells(const NodesDev &nd, const TickDev &t) { return (t.n_active - t.first_new) * cdiv(nd.pos_span, kDnChunk); }

} // namespace rm
