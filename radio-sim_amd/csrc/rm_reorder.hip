// rm_reorder.hip -- per-frame reorder to node-index order and the java.util.Random draw kernels
// (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math; overview at the top of rm_engine.h)
#include "rm_device.hpp"

#include <stdlib.h>

namespace rm {

// Sorted tables: the heard links of a frame sit unordered in the frame's segment of the A
// records.  One wave per frame ranks them by node index -- the order the reference's loop visits
// receivers in (UDGMRadioMedium.java:99) -- and writes them to their final, compact place.
// A segment of up to 64 links sits one per lane and is ranked with a readlane loop, longer ones by
// counting through memory.  MODE 1: the scan of the per-frame heard counts is redone in every
// workgroup (LDS); MODE 2: slot_off comes from k_scan_counts.
// cwl: log2 of the run of CONSECUTIVE frames a wave takes at a time (its runs are gridDim.x * 4 runs apart).  0 is a frame per
// step: the waves' frames interleave, every wave gets the same share of a tick whose frames are wave-sized jobs.  Where the
// lanes take a frame each (three links per frame, thousands of frames), neighbouring lanes with neighbouring frames read
// neighbouring segments and write neighbouring records: 64 frames G apart were 64 lines per load and per store instruction
// (configs[3]: 888 MB of counter traffic per 128 ticks for 45 MB of records).
template <bool STOCH, bool SINR, int MODE>
RM_D void reorder_body(const ModelDev &m, const TickDev &t, const int cwl)
{
    __shared__ uint32_t s_off[scan_lds(MODE)];
    __shared__ uint32_t s_wave[4];
    const bool publisher = blockIdx.x == 0;
    const int lane = threadIdx.x & 63;
    const int n_new = t.n_active - t.first_new;
    constexpr bool kRegScan = (MODE == 3 || MODE == 4);
    SmallCounts<scan_per(MODE)> pre{};
    if (kRegScan) pre = small_scan_load<scan_per(MODE)>(t.cursor, t.n_cnt);

    // the first frame of this wave: its records are requested before the scan below, so that the
    // scan's round trip and the records' overlap
    const int W = blockIdx.x * 4 + wave_index();
    const int G = int(gridDim.x) * 4;
    // the wave's k-th frame: run k >> cwl of its runs, place k & (2^cwl - 1) in it (increasing in k)
    auto frame_of = [&](const int k) -> int64_t { return ((int64_t(W) + int64_t(k >> cwl) * G) << cwl) + int64_t(k & ((1 << cwl) - 1)); };
    const int64_t q0 = frame_of(0);
    uint32_t src0 = 0, len = 0;
    int mine = 0x7fffffff, in_e = 0;
    double in_rssi = 0.0, in_prob = 1.0;
    uint8_t v = 0;
    if (q0 < n_new) {
        src0 = uniform_u(t.seg_off[int(q0) + t.shift]);
        len = uniform_u(t.cursor[int(q0) + t.shift]);
        if (uint32_t(lane) < len) {
            const uint32_t o = src0 + lane;
            mine = t.a_dst[o];
            in_rssi = t.a_rssi[o];
            v = t.a_verdict[o];
            if (STOCH) in_prob = t.a_prob[o];
            if (SINR && !t.seg_ordered && t.acc_lo == nullptr) in_e = t.a_e[o]; // (ordered segments and summed ticks carry sinr and verdict themselves)
        }
    }

    if (MODE == 1 || kRegScan) {
        uint32_t vmax = 0;
        const uint32_t total = kRegScan
                                   ? small_scan(pre, t.n_cnt, s_off, s_wave, publisher ? t.slot_off : nullptr, publisher ? &vmax : nullptr)
                                   : block_scan_counts(t.cursor, t.n_cnt, s_off, s_wave, publisher ? t.slot_off : nullptr,
                                                       publisher ? &vmax : nullptr);
        if (publisher && threadIdx.x == 0) {
            t.out_count[0] = total < t.cap ? total : t.cap;
            t.out_count[1] = (total > t.cap || t.stage_count[1] != 0u) ? 1u : 0u;
            t.out_count[2] = total;
            t.out_count[3] = vmax;
        }
        // a rank's frame list (k_rank_frames): the packets' offsets by GLOBAL number -- an unlisted packet has no links here
        if (publisher && t.fl_lb != nullptr)
            for (int g = int(threadIdx.x); g <= t.n_pub; g += int(blockDim.x)) t.pub_off[g] = s_off[t.shift + int(t.fl_lb[g])];
    } else if (publisher && threadIdx.x == 0) {
        const uint32_t total = t.slot_off[t.n_cnt];
        t.out_count[0] = total < t.cap ? total : t.cap;
        t.out_count[1] = (total > t.cap || t.stage_count[1] != 0u) ? 1u : 0u;
        t.out_count[2] = total;
    }

    // The wave's frames frame_of(0), frame_of(1), ...  With the scan in LDS a frame's count is the difference of two offsets
    // there, so the frames are looked at 64 at a time, one per lane: one vector load brings the segment offsets of those
    // that heard anything, and only those are walked -- a receiver partition hears nothing of most frames, and a chain of
    // two dependent global loads per frame (offset, count) was most of this stage's time on a rank's share of a tick.
    constexpr bool kLdsCounts = (MODE == 1 || kRegScan);
    for (int ib = 0;; ib += 64) { // wave-uniform
        if (frame_of(ib) >= n_new) break;
        uint32_t my_len = 0, my_src = 0;
        uint64_t todo = ~0ull;
        if (kLdsCounts) {
            const int64_t qi = frame_of(ib + lane);
            if (qi < n_new) {
                const int sl = int(qi) + t.shift;
                my_len = s_off[sl + 1] - s_off[sl];
                if (my_len) my_src = t.seg_off[sl];
            }
            // A frame with a handful of heard links (sixteen channels: three per frame) is its LANE's own business: the lane ranks its
            // frame's links in registers and writes them -- 64 frames at a time and their loads all in flight together, where a
            // wave per frame went through the same dependent round trips for three live lanes of 64 (configs[3]: 5000 frames
            // per tick, 445 us of this stage per 128 ticks).  Longer frames keep the wave.
            constexpr uint32_t kLaneMax = 8;
            const bool lane_frames = !t.seg_ordered && !(SINR && t.air.pool != nullptr);
            if (lane_frames && my_len != 0u && my_len <= kLaneMax) {
                const int q = int(qi);
                const int slot = q + t.shift;
                const uint32_t dst0 = s_off[slot];
                const int q_pub = t.fl_map ? t.fl_map[q] : q;
                int dd[kLaneMax];
#pragma unroll
                for (uint32_t k = 0; k < kLaneMax; ++k) dd[k] = (k < my_len) ? t.a_dst[my_src + k] : 0x7fffffff;
#pragma unroll
                for (uint32_t k = 0; k < kLaneMax; ++k) {
                    if (k >= my_len) break;
                    uint32_t rank = 0;
#pragma unroll
                    for (uint32_t j = 0; j < kLaneMax; ++j) rank += (dd[j] < dd[k]) ? 1u : 0u;
                    const uint32_t o = my_src + k, d = dst0 + rank;
                    if (d >= t.cap) continue;
                    t.out_pkt[d] = q_pub;
                    t.out_dst[d] = dd[k];
                    t.out_rssi[d] = t.a_rssi[o];
                    uint8_t vv = t.a_verdict[o];
                    if (SINR && t.acc_lo != nullptr) {
                        t.out_sinr[d] = t.a_sinr[o];
                    } else if (SINR) {
                        const int e = t.a_e[o];
                        t.out_sinr[d] = t.st_sinr[e];
                        if (t.st_coll[e]) vv = RM_INTERFERED;
                    }
                    t.out_verdict[d] = vv;
                    if (STOCH) t.out_prob[d] = t.a_prob[o];
                }
                my_len = 0u; // done
            }
            todo = ballot64(my_len != 0u);
        }
    while (todo) { // wave-uniform
        const int i = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t q64 = frame_of(ib + i);
        if (q64 >= n_new) break; // (MODE 2 walks every frame of the block)
        const int q = int(q64);
        const int slot = q + t.shift;
        const int q_pub = t.fl_map ? uniform_i(t.fl_map[q]) : q; // the packet's number in the result (a rank's frame list: global)
        if (kLdsCounts) {
            src0 = uint32_t(__builtin_amdgcn_readlane(int(my_src), i));
            len = uint32_t(__builtin_amdgcn_readlane(int(my_len), i));
        } else if (int64_t(q) != q0) {
            src0 = uniform_u(t.seg_off[slot]);
            len = uniform_u(t.cursor[slot]);
        }
        const uint32_t dst0 = uniform_u(kLdsCounts ? s_off[slot] : t.slot_off[slot]);
        for (uint32_t c0 = 0; c0 < len; c0 += 64) {
            const uint32_t o = src0 + c0 + lane;
            const bool valid = c0 + lane < len;
            if (int64_t(q) != q0 || c0 != 0) { // everything but the prefetched first chunk
                mine = valid ? t.a_dst[o] : 0x7fffffff;
                in_rssi = valid ? t.a_rssi[o] : 0.0;
                v = valid ? t.a_verdict[o] : uint8_t(0);
                in_prob = (STOCH && valid) ? t.a_prob[o] : 1.0;
                in_e = (SINR && valid && !t.seg_ordered && t.acc_lo == nullptr) ? t.a_e[o] : 0;
            }
            SinrOut so = {0.0, false};
            if (SINR && t.acc_lo != nullptr) { // interference summed per receiver: k_sinr_acc has left sinr and verdict in the A record
                if (valid) so.sinr = t.a_sinr[o];
            } else if (SINR && t.seg_ordered) { // the one-launch tick: k_sinr_frames / k_sinr_scan have written sinr and verdict into the segment
                if (valid) so.sinr = t.a_sinr[o];
            } else if (SINR && t.air.pool != nullptr && valid) { // the lists that live across ticks: the walk happens here, one lane per heard link
                const rm_tx_record &w = t.tx[t.first_new + q];
                so = air_sinr(m, t, t.st_dst[in_e], t.st_next[in_e], w.start_us, w.air_us, in_rssi);
            }
            uint32_t rank = 0;
            if (t.seg_ordered) {
                rank = c0 + lane;
            } else if (len <= 64) {
                for (uint32_t i = 0; i < len; ++i) rank += (__builtin_amdgcn_readlane(mine, int(i)) < mine) ? 1u : 0u;
            } else if (valid) {
                for (uint32_t k = 0; k < len; ++k) rank += (t.a_dst[src0 + k] < mine) ? 1u : 0u;
            }
            const uint32_t d = dst0 + rank;
            if (valid && d < t.cap) {
                t.out_pkt[d] = q_pub;
                t.out_dst[d] = mine;
                t.out_rssi[d] = in_rssi;
                uint8_t vv = v;
                if (SINR && (t.seg_ordered || t.air.pool != nullptr || t.acc_lo != nullptr)) {
                    t.out_sinr[d] = so.sinr;
                    if (so.collided) vv = RM_INTERFERED;
                } else if (SINR) {
                    t.out_sinr[d] = t.st_sinr[in_e];
                    if (t.st_coll[in_e]) vv = RM_INTERFERED;
                } // no SINR extension: the record carries no sinr (the array is not even allocated; readers give 0)
                t.out_verdict[d] = vv;
                if (STOCH) t.out_prob[d] = in_prob;
            }
        }
    }
    }
    if (!STOCH && t.fl_map == nullptr) write_pkt_interference(m, t, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x); // (a frame list: k_rank_frames wrote them, by global number)
    if (SINR && t.air.pool != nullptr && !t.seg_ordered && publisher) air_end(t);
}

template <bool STOCH, bool SINR, int MODE>
__global__ void __launch_bounds__(256) k_reorder(ModelDev m, TickDev t)
{
    reorder_body<STOCH, SINR, MODE>(m, t, 0);
}

template <bool STOCH, int SCAN, bool SINR = false>
__global__ void __launch_bounds__(256) k_reorder_batch(const ModelDev m, const TickDev *__restrict__ ticks, const int cwl)
{
    reorder_body<STOCH, SINR, SCAN>(m, ticks[blockIdx.z], cwl);
}

// ============================================================================ Java-RNG draws

constexpr int kScanTile = 2048; // 256 threads x 8

RM_D uint32_t draw_flag(const TickDev &t, uint32_t pos)
{
    return (t.out_verdict[pos] == 0 && t.out_prob[pos] < 1.0) ? 1u : 0u;
}

// (the tile kernels walk the tiles with a grid stride: the grid is sized for the records there are, not
// for the link capacity, and the batched variants share it between the ticks of a batch)
RM_D void draw_tile_sums_body(const TickDev &t)
{
    __shared__ uint32_t s_part[4];
    const uint32_t n = t.out_count[0];
    for (uint32_t tile = blockIdx.x; tile * kScanTile < n; tile += gridDim.x) { // block-uniform
        const uint32_t base = tile * kScanTile;
        uint32_t v = 0;
        for (int i = 0; i < 8; ++i) {
            const uint32_t pos = base + i * 256 + threadIdx.x;
            if (pos < n) v += draw_flag(t, pos);
        }
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) t.scan_block[tile] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_draw_tile_sums(TickDev t) { draw_tile_sums_body(t); }
__global__ void __launch_bounds__(256) k_draw_tile_sums_batch(const TickDev *__restrict__ ticks) { draw_tile_sums_body(ticks[blockIdx.z]); }

RM_D void draw_tile_scan_body(const TickDev &t)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t n = t.out_count[0];
    const int n_tiles = int((n + kScanTile - 1) / kScanTile);
    uint32_t carry = 0;
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < n_tiles) ? t.scan_block[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
        if (i < n_tiles) t.scan_block[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) t.draw_scan[n] = carry;
}

__global__ void __launch_bounds__(1024) k_draw_tile_scan(TickDev t) { draw_tile_scan_body(t); }
__global__ void __launch_bounds__(1024) k_draw_tile_scan_batch(const TickDev *__restrict__ ticks) { draw_tile_scan_body(ticks[blockIdx.z]); }

RM_D void draw_scan_body(const TickDev &t)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t n = t.out_count[0];
    for (uint32_t tile = blockIdx.x; tile * kScanTile < n; tile += gridDim.x) { // block-uniform
    const uint32_t base = tile * kScanTile;
    // thread owns 8 consecutive positions
    const uint32_t p0 = base + threadIdx.x * 8;
    uint32_t f[8];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f[i] = (p0 + i < n) ? draw_flag(t, p0 + i) : 0u;
        sum += f[i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(sum, lane);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t off = t.scan_block[tile] + inc - sum;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (p0 + i < n) t.draw_scan[p0 + i] = off;
        off += f[i];
    }
    __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_draw_scan(TickDev t) { draw_scan_body(t); }
__global__ void __launch_bounds__(256) k_draw_scan_batch(const TickDev *__restrict__ ticks) { draw_scan_body(ticks[blockIdx.z]); }

// per-packet number of receiver draws this rank would consume (if the packet's Tx does not fail)
RM_D void pkt_draw_counts_body(const TickDev &t)
{
    const int n_new = t.n_active - t.first_new;
    const uint32_t n = t.out_count[0];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_new) return;
    const uint32_t b = min(t.slot_off[q + t.shift], n);
    const uint32_t e = min(t.slot_off[q + t.shift + 1], n);
    t.pkt_draw_cnt[q] = t.draw_scan[e] - t.draw_scan[b];
}

__global__ void __launch_bounds__(256) k_pkt_draw_counts(TickDev t) { pkt_draw_counts_body(t); }
__global__ void __launch_bounds__(256) k_pkt_draw_counts_batch(const TickDev *__restrict__ ticks) { pkt_draw_counts_body(ticks[blockIdx.z]); }

// The only sequential part: the shared generator is consumed packet after packet
// (Simulator.getRandom(); UDGMRadioMedium.java:85-92,106).  One workgroup: the per-packet jump
// maps are built in parallel, then one lane walks the packets.  Receiver-sharded ranks all run
// the same chain on the all-gathered per-(rank, packet) draw counts `all_cnt[world][n_new]`: a
// packet's receivers are visited in node order = rank order, so this rank's first draw of packet q
// comes after the draws of the lower ranks, and the generator moves on by the sum over all ranks.
// The packet walk of java.util.Random.  A packet's step is
//   sure success (txSuccess >= 1): s -> R(s),   R = the jump over its receivers' draws;
//   sure failure (txSuccess <= 0): s -> s;
//   a Tx draw (0 < txSuccess < 1): s1 = J(s) (one nextDouble = two LCG steps), and s -> R(s1) if the
//   drawn value is <= txSuccess, else s -> s1 (UDGMRadioMedium.java:85-92, :106).
// Only the last case depends on the state, and it fails rarely.  So the walk is speculated a wave at a
// time: assuming every Tx draw succeeds, each packet is an affine map mod 2^48 and the state before
// packet l of a chunk of 64 is P_l(v), P_l = the exclusive prefix composition (a wave scan, done
// for all 16 chunks in parallel).  One wave then checks a chunk's 64 draws at once; at the first
// packet f whose draw fails the packets before it are final, f itself becomes s -> J(s_f), and
// the rest of the chunk is re-based without a new scan: v' = P_{f+1}^{-1}(J(s_f)).  The comparison
// nextDouble() > txSuccess is done on integers: nextDouble = k * 2^-53 with k < 2^53, so it is
// k > floor(txSuccess * 2^53) (the scaling by 2^53 is exact).
struct Affine {
    uint64_t a, c;
};
RM_D Affine affine_after(Affine later, Affine earlier) // later o earlier
{
    Affine r;
    r.a = (later.a * earlier.a) & kLcgMask;
    r.c = (later.a * earlier.c + later.c) & kLcgMask;
    return r;
}
RM_D uint64_t inv_mod_2_48(uint64_t a) // a odd
{
    uint64_t x = a; // 3 correct bits
    for (int i = 0; i < 4; ++i) x = x * (2ull - a * x); // 6, 12, 24, 48
    return x & kLcgMask;
}
RM_D uint64_t shfl_u64(uint64_t v, int src)
{
    const uint32_t lo = uint32_t(__shfl(int(uint32_t(v)), src)), hi = uint32_t(__shfl(int(uint32_t(v >> 32)), src));
    return (uint64_t(hi) << 32) | lo;
}
RM_D uint64_t shfl_up_u64(uint64_t v, int d)
{
    const uint32_t lo = uint32_t(__shfl_up(int(uint32_t(v)), d)), hi = uint32_t(__shfl_up(int(uint32_t(v >> 32)), d));
    return (uint64_t(hi) << 32) | lo;
}
RM_D uint64_t first_lane_u64(uint64_t v)
{
    return (uint64_t(uniform_u(uint32_t(v >> 32))) << 32) | uniform_u(uint32_t(v));
}
RM_D uint64_t read_lane_u64(uint64_t v, int src) // src wave-uniform
{
    const uint32_t lo = uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v)), src));
    const uint32_t hi = uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v >> 32)), src));
    return (uint64_t(hi) << 32) | lo;
}

RM_D void rng_chain_body(const ModelDev &m, const TickDev &t, const uint32_t *all_cnt, int world, int rank)
{
    __shared__ uint64_t s_pa[1024], s_pc[1024];   // exclusive prefix map of the packet inside its chunk
    __shared__ uint64_t s_pinv[1024];             // s_pa^-1 mod 2^48 (re-basing after a failed draw)
    __shared__ uint64_t s_thr[1024];              // floor(txSuccess * 2^53) of a drawing packet
    __shared__ uint64_t s_ab[1024], s_cb[1024];   // jump over the lower ranks' receiver draws
    __shared__ uint64_t s_start[1024];            // state where the packet's receiver draws begin
    __shared__ uint64_t s_ta[16], s_tc[16];       // a chunk's whole map
    __shared__ uint8_t s_kind[1024], s_fail[1024];
    __shared__ uint64_t s_state;
    enum { kSure = 0, kDraw = 1, kDead = 2 };
    const int n_new = t.n_active - t.first_new;
    const int lane = threadIdx.x & 63, wave = wave_index();
    Affine two; // one nextDouble
    two.a = (kLcgA * kLcgA) & kLcgMask;
    two.c = ((kLcgA + 1) * kLcgC) & kLcgMask;
    if (threadIdx.x == 0) s_state = *t.rng_state & kLcgMask;
    __syncthreads();
    for (int base = 0; base < n_new; base += 1024) {
        // (1) every packet's maps, every chunk's prefix composition
        const int q = base + threadIdx.x;
        Affine mq = {1ull, 0ull};
        uint8_t kind = kSure;
        uint64_t thr = 0;
        Affine before = {1ull, 0ull};
        if (q < n_new) {
            uint64_t total = 0, nb = 0;
            if (all_cnt) {
                for (int r = 0; r < world; ++r) {
                    const uint32_t v = all_cnt[size_t(r) * n_new + q];
                    if (r < rank) nb += v;
                    total += v;
                }
            } else {
                total = t.pkt_draw_cnt[q];
            }
            Affine recv;
            lcg_jump_map(2ull * total, recv.a, recv.c);
            lcg_jump_map(2ull * nb, before.a, before.c);
            const double txs = tx_success(m, t.tx[t.first_new + q]);
            if (txs <= 0.0) {
                kind = kDead;
            } else if (txs < 1.0) {
                kind = kDraw;
                thr = uint64_t(txs * 0x1.0p53);
                mq = affine_after(recv, two);
            } else {
                mq = recv;
            }
        }
        Affine inc = mq; // inclusive prefix: packets 0..lane of the chunk, applied in packet order
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            Affine prev;
            prev.a = shfl_up_u64(inc.a, d);
            prev.c = shfl_up_u64(inc.c, d);
            if (lane >= d) inc = affine_after(inc, prev);
        }
        Affine exc; // exclusive
        exc.a = shfl_up_u64(inc.a, 1);
        exc.c = shfl_up_u64(inc.c, 1);
        if (lane == 0) exc = {1ull, 0ull};
        s_pa[threadIdx.x] = exc.a;
        s_pc[threadIdx.x] = exc.c;
        s_pinv[threadIdx.x] = inv_mod_2_48(exc.a);
        s_thr[threadIdx.x] = thr;
        s_ab[threadIdx.x] = before.a;
        s_cb[threadIdx.x] = before.c;
        s_kind[threadIdx.x] = kind;
        if (lane == 63) {
            s_ta[wave] = inc.a;
            s_tc[wave] = inc.c;
        }
        __syncthreads();
        // (2) one wave walks the chunks
        if (wave == 0) {
            uint64_t s0 = s_state; // the true state before the chunk's first packet
            const int chunks = (min(1024, n_new - base) + 63) >> 6;
            for (int c = 0; c < chunks; ++c) {
                const int idx = c * 64 + lane;
                const uint64_t pa = s_pa[idx], pc = s_pc[idx], pinv = s_pinv[idx], th = s_thr[idx];
                const int kd = s_kind[idx];
                uint64_t v = s0; // virtual start: the state before packet l is P_l(v) for the lanes >= lo
                int lo = 0;
                bool done = false;
                while (!done) { // wave-uniform; at most 64 rounds
                    const uint64_t sl = (pa * v + pc) & kLcgMask;
                    const uint64_t s1 = (sl * kLcgA + kLcgC) & kLcgMask;
                    const uint64_t s2 = (s1 * kLcgA + kLcgC) & kLcgMask;
                    const uint64_t k = ((s1 >> 22) << 27) + (s2 >> 21);
                    const bool fails = (lane >= lo) && (kd == kDraw) && (k > th);
                    const uint64_t fm = ballot64(fails);
                    const int f = fm ? (__ffsll((long long)fm) - 1) : 64; // wave-uniform
                    if (lane >= lo && lane <= f) { // final now
                        s_start[idx] = (kd == kDraw) ? s2 : sl;
                        s_fail[idx] = (kd == kDead || lane == f) ? 1 : 0;
                    }
                    if (f >= 64) {
                        s0 = (first_lane_u64(s_ta[c]) * v + first_lane_u64(s_tc[c])) & kLcgMask;
                        done = true;
                    } else {
                        const uint64_t y = read_lane_u64(s2, f); // the failed packet consumed its Tx draw only
                        if (f == 63) {
                            s0 = y;
                            done = true;
                        } else { // re-base the rest of the chunk: P_{f+1}(v') = y
                            v = (read_lane_u64(pinv, f + 1) * ((y - read_lane_u64(pc, f + 1)) & kLcgMask)) & kLcgMask;
                            lo = f + 1;
                        }
                    }
                }
            }
            if (lane == 0) s_state = s0;
        }
        __syncthreads();
        // (3) every packet's Tx-failure flag and the state its receivers' draws start from on this rank
        if (q < n_new) {
            t.pkt_interference[q] = s_fail[threadIdx.x];
            t.pkt_rng[q] = (s_ab[threadIdx.x] * s_start[threadIdx.x] + s_cb[threadIdx.x]) & kLcgMask;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *t.rng_state = s_state;
    __syncthreads(); // the generator state is in memory before a next tick of the same launch reads it
}

__global__ void __launch_bounds__(1024) k_rng_chain(ModelDev m, TickDev t, const uint32_t *all_cnt, int world, int rank)
{
    rng_chain_body(m, t, all_cnt, world, rank);
}

// the ticks of a batch share the generator: one workgroup walks them in slot order
__global__ void __launch_bounds__(1024) k_rng_chain_batch(const ModelDev m, const TickDev *__restrict__ ticks, int n)
{
    for (int b = 0; b < n; ++b) rng_chain_body(m, ticks[b], nullptr, 1, 0);
}

RM_D void apply_draws_body(const TickDev &t)
{
    const uint32_t n = t.out_count[0];
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += stride) {
        const int q = t.out_pkt[pos];
        uint8_t v = t.out_verdict[pos];
        if (t.pkt_interference[q]) {
            v = RM_INTERFERED; // UDGMRadioMedium.java:106: no draw once the Tx failed
        } else if (v == 0) {
            const double p = t.out_prob[pos];
            if (p < 1.0) {
                const uint32_t first = min(t.slot_off[q + t.shift], n);
                const uint32_t k = t.draw_scan[pos] - t.draw_scan[first];
                uint64_t A, C;
                lcg_jump_map(2ull * k, A, C);
                uint64_t s = (A * t.pkt_rng[q] + C) & kLcgMask;
                v = (lcg_next_double(s) > p) ? RM_INTERFERED : RM_DELIVERED;
            } else {
                v = RM_DELIVERED;
            }
        }
        t.out_verdict[pos] = v;
    }
}

__global__ void __launch_bounds__(256) k_apply_draws(TickDev t) { apply_draws_body(t); }
__global__ void __launch_bounds__(256) k_apply_draws_batch(const TickDev *__restrict__ ticks) { apply_draws_body(ticks[blockIdx.z]); }

// ---- draws under a SPATIAL receiver partition ------------------------------------------------------------------
// The reference visits a packet's receivers in node order (UDGMRadioMedium.java:99) and draws where :106 says so.  With
// receivers partitioned by region the ranks' node sets interleave, so a rank cannot place its draws from the other
// ranks' COUNTS alone: every rank also publishes, packet-major, the node index of each of its links that will draw
// (k_draw_nodes), the lists are exchanged, and a link's place among its packet's draws is the number of listed nodes
// below it over all ranks (a binary search per rank: every list is ascending inside a packet).
__global__ void __launch_bounds__(256) k_draw_nodes(TickDev t, int32_t *__restrict__ dn)
{
    const uint32_t n = t.out_count[0];
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += stride)
        if (draw_flag(t, pos)) dn[t.draw_scan[pos]] = t.out_dst[pos];
}

// all_off[r][q] = first entry of packet q in rank r's node list = exclusive scan of all_cnt[r][.] (one workgroup per rank)
__global__ void __launch_bounds__(1024) k_draw_offsets(const uint32_t *__restrict__ all_cnt, int n_new, uint32_t *__restrict__ all_off)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t *cnt = all_cnt + size_t(blockIdx.x) * n_new;
    uint32_t *off = all_off + size_t(blockIdx.x) * n_new;
    uint32_t carry = 0;
    for (int base = 0; base < n_new; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < n_new) ? cnt[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
        if (i < n_new) off[i] = carry + ex;
        carry += total;
    }
}

__global__ void __launch_bounds__(256)
k_apply_draws_nodes(TickDev t, const uint32_t *__restrict__ all_cnt, const uint32_t *__restrict__ all_off,
                    const int32_t *__restrict__ all_nodes, uint32_t stride, int world)
{
    const uint32_t n = t.out_count[0];
    const int n_new = t.n_active - t.first_new;
    const uint32_t step = gridDim.x * blockDim.x;
    for (uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += step) {
        const int q = t.out_pkt[pos];
        uint8_t v = t.out_verdict[pos];
        if (t.pkt_interference[q]) {
            v = RM_INTERFERED; // UDGMRadioMedium.java:106: no draw once the Tx failed
        } else if (v == 0) {
            const double p = t.out_prob[pos];
            if (p < 1.0) {
                const int node = t.out_dst[pos];
                uint32_t k = 0; // drawing links of this packet with a smaller node index, over all ranks
                for (int r = 0; r < world; ++r) {
                    const int32_t *lst = all_nodes + size_t(r) * stride + all_off[size_t(r) * n_new + q];
                    uint32_t lo = 0, hi = all_cnt[size_t(r) * n_new + q];
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (lst[mid] < node) lo = mid + 1; else hi = mid;
                    }
                    k += lo;
                }
                uint64_t A, C;
                lcg_jump_map(2ull * k, A, C);
                uint64_t s = (A * t.pkt_rng[q] + C) & kLcgMask; // pkt_rng: where the packet's receiver draws begin (chain with rank 0)
                v = (lcg_next_double(s) > p) ? RM_INTERFERED : RM_DELIVERED;
            } else {
                v = RM_DELIVERED;
            }
        }
        t.out_verdict[pos] = v;
    }
}

// ============================================================================ launchers

hipError_t launch_draw_nodes(hipStream_t s, const TickDev &t, int32_t *dev_nodes)
{
    RM_KLAUNCH(k_draw_nodes, dim3(256), dim3(256), 0, s, t, dev_nodes);
    return hipGetLastError();
}

hipError_t launch_draws_apply_nodes(hipStream_t s, const ModelDev &m, const TickDev &t, const uint32_t *all_cnt, uint32_t *all_off,
                                    const int32_t *all_nodes, uint32_t stride, int world)
{
    const int n_new = t.n_active - t.first_new;
    if (n_new > 0) RM_KLAUNCH(k_draw_offsets, dim3(world), dim3(1024), 0, s, all_cnt, n_new, all_off);
    RM_KLAUNCH(k_rng_chain, dim3(1), dim3(1024), 0, s, m, t, all_cnt, world, 0);
    RM_KLAUNCH(k_apply_draws_nodes, dim3(256), dim3(256), 0, s, t, all_cnt, all_off, all_nodes, stride, world);
    return hipGetLastError();
}

// sorted tables only
hipError_t launch_reorder(hipStream_t s, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg)
{
    const int n_new = t.n_active - t.first_new;
    const dim3 grid(max(1, min(2048, (n_new + 3) / 4))), block(256);
    const bool sinr = (m.kind == RM_MODEL_LOGDIST) && (m.flags & RM_LD_SINR);
    const int mode = scan_variant(t.n_cnt);
#define RM_RE(ST, SI, MO) RM_KLAUNCH((k_reorder<ST, SI, MO>), grid, block, 0, s, m, t)
    if (mode == 3) {
        if (cfg.stochastic) { if (sinr) RM_RE(true, true, 3); else RM_RE(true, false, 3); }
        else { if (sinr) RM_RE(false, true, 3); else RM_RE(false, false, 3); }
    } else if (mode == 4) {
        if (cfg.stochastic) { if (sinr) RM_RE(true, true, 4); else RM_RE(true, false, 4); }
        else { if (sinr) RM_RE(false, true, 4); else RM_RE(false, false, 4); }
    } else if (mode == 1) {
        if (cfg.stochastic) { if (sinr) RM_RE(true, true, 1); else RM_RE(true, false, 1); }
        else { if (sinr) RM_RE(false, true, 1); else RM_RE(false, false, 1); }
    } else {
        if (cfg.stochastic) { if (sinr) RM_RE(true, true, 2); else RM_RE(true, false, 2); }
        else { if (sinr) RM_RE(false, true, 2); else RM_RE(false, false, 2); }
    }
#undef RM_RE
    return hipGetLastError();
}

// rm_batch_*, stage 2
hipError_t launch_reorder_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n, const TickDev *b,
                                const LaunchCfg &cfg)
{
    const int scan = batch_scan_variant(ticks, n);
    int max_new = 0;
    for (int i = 0; i < n; ++i) max_new = max(max_new, ticks[i].n_active - ticks[i].first_new);
    // two frames per wave (every workgroup redoes the scan of the per-frame counts first: fewer, longer
    // workgroups); a receiver partition hears 1/share of a frame's links, so its waves take more
    // ... and every workgroup redoes the scan over all frames of its tick: with thousands of frames per tick
    // fewer workgroups do it (configs[3], 5000 frames: 36.6 -> 33.8 us per tick)
    // A receiver partition hears 1/share of the links and nothing at all of most frames: far more frames per wave (a rank's
    // share of an 8-GPU tick: configs[2] 16 / 32 / 64 frames per wave 0.536 / 0.513 / 0.505 us per tick, configs[3] with its
    // 5000 frames 32 / 64 / 128 / 256: 4.98 / 4.58 / 4.37 / 4.27 us).
    const int share = (nd.n_rx > 0 && nd.n_rx < nd.n) ? max(1, nd.n / nd.n_rx) : 1;
    int fpw = max(2, min(32, max_new / 256));
    if (share > 1) fpw = min(256, fpw * 4 * share);
    // (SINR ticks summed per receiver -- sixteen channels, three links per frame: the lanes take the frames one each, a workgroup
    // walks 256 frames per pass and every workgroup redoes the scan over all the tick's frames: few, long workgroups.
    // configs[3]: 64 / 256 frames per wave 12.3 / 12.2 us per tick against 12.8 with 19)
    if (ticks[0].acc_lo != nullptr && share == 1) fpw = 256;
    // (a rank's frame list: every listed frame can have links and is a wave's job -- 96 frames per wave left an interior rank's ~250
    // listed frames to ONE workgroup per tick: 117-121 us per 512 ticks of configs[2] / [4] where the corner rank took 76; 16: 61-65)
    if (ticks[0].n_pub > 0 && share > 1 && ticks[0].acc_lo == nullptr) fpw = 16;
    if (const char *e = getenv("RM_FPW")) fpw = max(1, atoi(e));
    // (a rank's frame list: the device walks the listed frames only -- about 1/share of them and a halo; the grid only has to be
    // large enough to be busy, every wave strides over whatever frames there are)
    const int walk = (ticks[0].n_pub > 0 && share > 1) ? max(64, min(max_new, max_new * 3 / share)) : max_new;
    const dim3 grid(max(1, min(2048, cdiv(walk, 4 * fpw))), 1, n), block(256);
    // runs of 64 consecutive frames per wave where the lanes take a frame each (the ticks summed per receiver: sixteen channels,
    // three links per frame -- configs[3] 229 -> 131 us per 128 ticks, a rank's share 122 -> 78 per 512); single frames where a frame
    // is a wave's job: runs there leave a tick's few hundred listed frames to three waves (a rank's share of configs[2]: 78 -> 187 us
    // with runs of 64).  RM_REORDER_RUN: log2 of the run, tests.
    int cwl = (ticks[0].acc_lo != nullptr) ? 6 : 0;
    if (const char *e = getenv("RM_REORDER_RUN")) cwl = max(0, min(6, atoi(e)));
    if (m.kind == RM_MODEL_LOGDIST && (m.flags & RM_LD_SINR)) {
        if (scan == 3) RM_KLAUNCH((k_reorder_batch<false, 3, true>), grid, block, 0, s, m, b, cwl);
        else if (scan == 4) RM_KLAUNCH((k_reorder_batch<false, 4, true>), grid, block, 0, s, m, b, cwl);
        else RM_KLAUNCH((k_reorder_batch<false, 1, true>), grid, block, 0, s, m, b, cwl);
    } else if (cfg.stochastic) {
        if (scan == 3) RM_KLAUNCH((k_reorder_batch<true, 3>), grid, block, 0, s, m, b, cwl);
        else if (scan == 4) RM_KLAUNCH((k_reorder_batch<true, 4>), grid, block, 0, s, m, b, cwl);
        else RM_KLAUNCH((k_reorder_batch<true, 1>), grid, block, 0, s, m, b, cwl);
    } else {
        if (scan == 3) RM_KLAUNCH((k_reorder_batch<false, 3>), grid, block, 0, s, m, b, cwl);
        else if (scan == 4) RM_KLAUNCH((k_reorder_batch<false, 4>), grid, block, 0, s, m, b, cwl);
        else RM_KLAUNCH((k_reorder_batch<false, 1>), grid, block, 0, s, m, b, cwl);
    }
    return hipGetLastError();
}

// draws, part 1: which ordered records need a draw, and how many per packet
hipError_t launch_draws_scan(hipStream_t s, const TickDev &t)
{
    const int tiles = min(256, cdiv(int(t.cap), kScanTile)); // grid-stride over the tiles that hold records
    const int n_new = t.n_active - t.first_new;
    RM_KLAUNCH(k_draw_tile_sums, dim3(tiles), dim3(256), 0, s, t);
    RM_KLAUNCH(k_draw_tile_scan, dim3(1), dim3(1024), 0, s, t);
    RM_KLAUNCH(k_draw_scan, dim3(tiles), dim3(256), 0, s, t);
    RM_KLAUNCH(k_pkt_draw_counts, dim3(max(1, cdiv(n_new, 256))), dim3(256), 0, s, t);
    return hipGetLastError();
}

// draws, part 2: walk the generator over the packets, then every flagged record draws at its place
hipError_t launch_draws_apply(hipStream_t s, const ModelDev &m, const TickDev &t, const uint32_t *all_cnt, int world,
                              int rank)
{
    RM_KLAUNCH(k_rng_chain, dim3(1), dim3(1024), 0, s, m, t, all_cnt, world, rank);
    RM_KLAUNCH(k_apply_draws, dim3(256), dim3(256), 0, s, t);
    return hipGetLastError();
}

// rm_batch_*: the draw stage of all ticks of a batch in six launches; the generator is walked over the
// ticks in slot order inside ONE launch of k_rng_chain_batch (full table only: no per-rank exchange)
hipError_t launch_draws_batch(hipStream_t s, const ModelDev &m, const TickDev *ticks, int n, const TickDev *b)
{
    int max_new = 0;
    for (int i = 0; i < n; ++i) max_new = max(max_new, ticks[i].n_active - ticks[i].first_new);
    const int tiles = min(64, cdiv(int(ticks[0].cap), kScanTile));
    RM_KLAUNCH(k_draw_tile_sums_batch, dim3(tiles, 1, n), dim3(256), 0, s, b);
    RM_KLAUNCH(k_draw_tile_scan_batch, dim3(1, 1, n), dim3(1024), 0, s, b);
    RM_KLAUNCH(k_draw_scan_batch, dim3(tiles, 1, n), dim3(256), 0, s, b);
    RM_KLAUNCH(k_pkt_draw_counts_batch, dim3(max(1, cdiv(max_new, 256)), 1, n), dim3(256), 0, s, b);
    RM_KLAUNCH(k_rng_chain_batch, dim3(1), dim3(1024), 0, s, m, b, n);
    RM_KLAUNCH(k_apply_draws_batch, dim3(32, 1, n), dim3(256), 0, s, b);
    return hipGetLastError();
}


} // namespace rm
