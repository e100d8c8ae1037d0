// rm_airscan.hip -- the SINR medium's lone tick without per-receiver lists: a heard link finds its interferers among the
// frames on the air themselves (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math)
//
// The SINR extension (not reference behaviour -- the reference media have no interference model)
// sums, for every heard link of a new frame, the linear power of every co-channel frame on the air that overlaps it in
// time and reaches the interference floor at that receiver; a receiver that is itself on the air is deaf (half duplex).
// The list form (rm_device.hpp: air_link / air_sinr) keeps that set per RECEIVER: every frame leaves an entry in the
// list of every receiver it is significant at -- about ten times the receivers that hear it -- and a heard link walks
// its receiver's list.  At 1 M nodes both ends are random accesses to 130 MB of entries: 460 k atomic exchanges and
// 260 k dependent line fetches per tick of 1000 frames were 40 of that tick's 63 us.
//
// Here nothing is kept per receiver.  The frames on the air are few (thousands) and the links of one new frame all end
// near that frame.  The first launch (k_tick_frames, rm_tick.hip) evaluates the heard links as the medium without SINR
// does and indexes every frame on the air from scratch (scan_index, rm_device.hpp): a scan record (position, cut-off
// radius at the interference floor, channel), its cell of a 64 x 64 grid over the fp32 frame, and its place in its source
// node's chain of frames.  Then, one new frame per workgroup:
//   near     the frames in the grid cells within (radius the frame is heard in + largest radius on the air) of the new
//            frame -- or, when that is most of the grid, every frame on the air -- are tested: co-channel frames whose
//            reach touches the circle the new frame is heard in form the frame's NEAR list (a few dozen);
//   links    the heard links' receivers (node, engine position, record: three dependent round trips, under the ones
//            above); a receiver whose own chain holds a frame that overlaps in time is deaf (half duplex);
//   pairs    (heard link, near frame): the sweep's own conservative tests -- fp32 distance against the frame's cut-off,
//            the shadowed medium's link-hash table -- leave the few pairs that can matter; every wave keeps its own;
//   exact    ... and evaluates them with full lanes: eval_link, the same fp64 arithmetic the list form ran when it
//            inserted the entry; the linear powers are added in Q80 fixed point (order-independent) to the link's sum;
//   verdict  sinr = rssi - 10 log10(sum + noise), capture threshold, half duplex -- air_sinr's formula.
// Every frame is evaluated against the node table as it is now: nothing can go stale,
// so there is no rebuild, no ring to overrun, and moving nodes cost nothing.  Frames whose record says they have left
// the air (start + air <= t_begin) are skipped, like entries of the lists.
#include "rm_device.hpp"

#include <stdlib.h>

namespace rm {

constexpr int kScLinks = 256;   // heard links of the frame handled together (one per thread)
constexpr int kScNear = 1024;   // near frames of the new frame ...
constexpr int kScNearW = kScNear / 4; // ... of which every wave gathers its own quarter (more: the slow path)
constexpr int kScU = 6;         // grid slots per thread requested together
constexpr int kScBatch = 4;     // scan records per thread requested together when every frame is looked at
constexpr int kScSub = 128;     // near frames whose records sit in LDS during a pair phase
constexpr int kScPairsW = 512;  // pairs that passed the conservative tests, per wave, between two exact phases
constexpr int kScPU = 3;        // pairs per lane tested together
constexpr int kScSteps = 2;     // (slow path) scan steps between two looks at the near list's fill
static_assert(kScLinks <= 256 && kScSub <= 256, "a pair is kept as link << 8 | near frame in 16 bits");
static_assert(kSgK == 16, "a grid slot is cell << 4 | entry");
static_assert(kScLinks * kScSub <= (1 << 15), "pair indices are divided by multiplication");

// Diagnostic build only (make stamps; never the shipped library): s_memtime at the phase boundaries, written to the unused
// tail of the compact rssi array, below k_tick_frames' own stamps (tools/scan_stamps.py reads them).
#ifdef RM_STAMPS
#define RM_STAMP(k)                                                                                          \
    do {                                                                                                     \
        if (threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memtime();                                      \
    } while (0)
#else
#define RM_STAMP(k)                                                                                          \
    do {                                                                                                     \
    } while (0)
#endif

RM_D void lds_add_u128(unsigned long long *acc /*[2]: lo, hi*/, const U128 v)
{
    if ((v.lo | v.hi) == 0ull) return;
    const unsigned long long old = atomicAdd(&acc[0], (unsigned long long)v.lo);
    const unsigned long long carry = (old + v.lo < old) ? 1ull : 0ull; // (the low words' running sum is exact mod 2^64: so is the carry count)
    if (v.hi + carry) atomicAdd(&acc[1], (unsigned long long)(v.hi + carry));
}

template <bool SHADOW>
__global__ void __launch_bounds__(256, 4) k_sinr_scan(const NodesDev nd, const ModelDev m, const TickDev t, const ScanDev sd)
{
    __shared__ int s_dst[kScLinks];                  // the links' receivers: node index
    __shared__ double s_rx[kScLinks * 3];            // ... their positions
    __shared__ float4 s_rxf[kScLinks];               // ... in the fp32 frame (+ channel bits)
    __shared__ unsigned long long s_acc[kScLinks * 2]; // Q80 interference sum per link
    __shared__ uint32_t s_hd[kScLinks];              // half duplex
    __shared__ int s_near[kScNear];
    __shared__ double s_fx[kScSub], s_fy[kScSub], s_fz[kScSub], s_fp[kScSub]; // the near frames: position, power ...
    __shared__ int s_fch[kScSub];                    // ... channel (field by field: records of 64 bytes would put every lane on two banks)
    __shared__ float4 s_ff[kScSub];                  // pre-filter record of a near frame at the interference floor (w < 0: not a partner)
    __shared__ float s_inv[kScSub];
    __shared__ int s_fsrc[kScSub];                   // ... its source
    __shared__ uint16_t s_pairs[4 * kScPairsW];
    __shared__ uint32_t s_tbl[SHADOW ? kShadowBins : 1];
    __shared__ uint32_t s_nn, s_over, s_wn[4];
    __shared__ float s_rmax;

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    const int slot = blockIdx.x;
    // the other parity's index counters, for the next tick by scan (nobody reads them during this one)
    for (int i = blockIdx.x * blockDim.x + tid; i < kSgCells + 1 + kSgMax; i += gridDim.x * blockDim.x) sd.cnt_next[i] = 0u;
    const int n_new = t.n_active - t.first_new;
    const int q = slot - t.shift;
    if (q < 0 || q >= n_new) return;
#ifdef RM_STAMPS
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(t.out_rssi + (size_t(t.cap) - 16u * 1024u - 16u * (size_t(slot) + 1u)));
    if (threadIdx.x == 0) stamps[15] = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < 15; ++k) RM_STAMP(k);
#endif
    const uint32_t len = uniform_u(t.cursor[slot]);
    if (len == 0u) return;
    const uint32_t src0 = uniform_u(t.seg_off[slot]);
    const int abs_q = t.first_new + q;
    const rm_tx_record wq = t.tx[abs_q];
    const int64_t q_end = wq.start_us + wq.air_us;
    const int n_every = uniform_i(int(sd.cnt[kSgCells]));
    if (tid < kSgMax) { // the largest radius on the air
        float r = __uint_as_float(sd.cnt[kSgCells + 1 + tid]);
        for (int d = 32; d >= 1; d >>= 1) r = fmaxf(r, __shfl_xor(r, d));
        if (tid == 0) s_rmax = r;
    }
    if (SHADOW) s_tbl[tid] = m.shadow_tbl[tid]; // kBlock == kShadowBins
    if (tid == 0) s_nn = s_over = 0u;

    // the circle this frame is heard in: its cut-off at the sensitivity level, as k_tick_frames used it
    float4 fq;
    {
        ModelDev ms = m;
        ms.ld_level = m.ld_sens;
        double thr64;
        tx_prefilter(ms, wq, fq, thr64);
    }
    const float rq = (fq.w >= 0.f) ? ((fq.w < __builtin_inff()) ? sqrt_up(fq.w) : __builtin_inff()) : 0.f;
    // near: co-channel, and its reach touches that circle (both radii carry the fp32 frame's slack; the margin covers the
    // rounding of this test)
    auto is_near = [&](const int i, const float4 &v, const int ch) -> bool {
        if (!(i != abs_q && v.w >= 0.f && ch == wq.channel)) return false;
        const float reach = rq + v.w;
        return dist2_f32(v.x - fq.x, v.y - fq.y, v.z - fq.z) <= reach * reach * (1.0f + 1e-5f);
    };

    // The first chunk of links: requested now, used after the near list (their three dependent round trips run under its two).
    const bool have0 = uint32_t(tid) < len;
    int node0 = 0;
    double rssi0 = 0.0;
    if (have0) {
        node0 = t.a_dst[src0 + uint32_t(tid)];
        rssi0 = t.a_rssi[src0 + uint32_t(tid)];
    }
    __syncthreads();
    const float rmax = s_rmax;

    // ---- the near list: every wave fills its own quarter, its count in a scalar register (no atomics, no barriers)
    int my_n = 0;
    bool over = false;
    auto append = [&](const bool near, const int i) {
        const uint64_t hm = ballot64(near);
        const int cnt = int(__popcll(hm));
        if (my_n + cnt > kScNearW) over = true; // wave-uniform
        else {
            if (near) s_near[wave * kScNearW + my_n + int(lane_prefix(hm))] = i;
            my_n += cnt;
        }
    };
    // which cells can hold a near frame?  |dx| <= rq + (largest radius), and positions map to cells monotonically
    bool use_grid = rq < __builtin_inff() && rmax < __builtin_inff();
    int cx0 = 0, cy0 = 0, gw = 1, gh = 1;
    if (use_grid) {
        const float reach = (rq + rmax) * (1.0f + 2e-5f) + 1e-3f / sd.inv;
        cx0 = sg_cell1(fq.x - reach, sd.half, sd.inv);
        cy0 = sg_cell1(fq.y - reach, sd.half, sd.inv);
        gw = sg_cell1(fq.x + reach, sd.half, sd.inv) - cx0 + 1;
        gh = sg_cell1(fq.y + reach, sd.half, sd.inv) - cy0 + 1;
        use_grid = gw * gh * kSgK <= 2 * t.n_active; // (otherwise looking at every frame is less work)
    }
    int pos0 = 0;
    unsigned long long self0 = 0ull;
    if (use_grid) {
        const int total = gw * gh * kSgK;
        const uint32_t inv_gw = (gw > 1) ? uint32_t((0x100000000ull + uint32_t(gw) - 1u) / uint32_t(gw)) : 0u;
        for (int s0 = 0; s0 < total; s0 += 256 * kScU) { // block-uniform
            uint32_t e_cnt[kScU];
            float4 v[kScU];
            int2 ci[kScU];
#pragma unroll
            for (int u = 0; u < kScU; ++u) { // a slot = an entry of a cell: the cell's count and the entry's scan record, requested together
                const int sl = min(s0 + u * 256 + tid, total - 1);
                const uint32_t cl = uint32_t(sl) >> 4;
                const uint32_t cyl = (gw > 1) ? __umulhi(cl, inv_gw) : cl; // cl / gw (exact: cl < 4096)
                const uint32_t cell = (uint32_t(cy0) + cyl) * uint32_t(kSgG) + uint32_t(cx0) + (cl - cyl * uint32_t(gw));
                e_cnt[u] = sd.cnt[cell];
                v[u] = sd.bucket_xyzr[cell * kSgK + (uint32_t(sl) & 15u)];
                ci[u] = sd.bucket_ci[cell * kSgK + (uint32_t(sl) & 15u)];
            }
            if (s0 == 0 && have0) { // (the links' second round trip)
                pos0 = engine_pos(nd, node0);
                self0 = sd.self_slot[node0];
            }
#pragma unroll
            for (int u = 0; u < kScU; ++u) {
                if (s0 + u * 256 >= total) break; // block-uniform
                const int sl = s0 + u * 256 + tid;
                const bool ok = sl < total && (uint32_t(sl) & 15u) < min(e_cnt[u], uint32_t(kSgK));
                append(ok && is_near(ci[u].y, v[u], ci[u].x), ci[u].y);
            }
        }
        for (int e0 = 0; e0 < n_every; e0 += 256) { // frames without a cell: no bound, outside the frame, or their cell was full
            const int e = e0 + tid;
            const int i = (e < n_every) ? int(sd.every[e]) : abs_q;
            append(is_near(i, sd.xyzr[i], sd.ch[i]), i);
        }
    } else {
        for (int i0 = 0; i0 < t.n_active; i0 += 256 * kScBatch) { // block-uniform
            float4 v[kScBatch];
            int ch[kScBatch];
#pragma unroll
            for (int k = 0; k < kScBatch; ++k) {
                const int i = min(i0 + k * 256 + tid, t.n_active - 1);
                v[k] = sd.xyzr[i];
                ch[k] = sd.ch[i];
            }
            if (i0 == 0 && have0) {
                pos0 = engine_pos(nd, node0);
                self0 = sd.self_slot[node0];
            }
#pragma unroll
            for (int k = 0; k < kScBatch; ++k) {
                if (i0 + k * 256 >= t.n_active) break; // block-uniform
                const int i = i0 + k * 256 + tid;
                append(i < t.n_active && is_near(i, v[k], ch[k]), i);
            }
        }
    }
    if (lane == 0) {
        s_wn[wave] = uint32_t(my_n);
        if (over) s_over = 1u;
    }
    RM_STAMP(1); // the near list

    // where the pair phase finds entry k of the near list: the waves' quarters one after the other, or (slow path) from the front
    int nb[4] = {0, 0, 0, 0}, np_[5] = {0, 0, 0, 0, 0};
    auto near_at = [&](const int k) -> int {
        const int w = (k >= np_[1] ? 1 : 0) + (k >= np_[2] ? 1 : 0) + (k >= np_[3] ? 1 : 0);
        return s_near[(w == 0 ? nb[0] : (w == 1 ? nb[1] : (w == 2 ? nb[2] : nb[3]))) + k - (w == 0 ? np_[0] : (w == 1 ? np_[1] : (w == 2 ? np_[2] : np_[3])))];
    };

    // (block-uniform) one pair phase: the near frames gathered so far against the chunk's links
    auto pairs_phase = [&](const int n_near, const int nl) {
        for (int c0 = 0; c0 < n_near; c0 += kScSub) {
            const int ns = min(kScSub, n_near - c0);
            __syncthreads(); // (the near frames of the phase before are done with; the links are in LDS)
            if (tid < kScSub) {
                float4 f = make_float4(0.f, 0.f, 0.f, -1.f);
                float inv = 0.f;
                int fs = -1;
                if (tid < ns) {
                    const rm_tx_record w = t.tx[near_at(c0 + tid)];
                    const int64_t w_end = w.start_us + w.air_us;
                    // on the air, and overlapping the new frame in time (air_sinr's test)
                    if (w.src >= 0 && w_end > t.air.t_begin && w.start_us < q_end && w_end > wq.start_us) {
                        double thr64;
                        tx_prefilter(m, w, f, thr64);
                        if (SHADOW && f.w > 0.f && f.w < __builtin_inff()) { // (the sweep's second-level filter, rm_tick.hip)
                            const float cut = __builtin_sqrtf(f.w);
                            if (1.01f * (2.0f * float(m.f32_slack)) / (0.15f * cut) + 1e-5f <= float(kShadowPad)) inv = float(kShadowBins) / f.w;
                        }
                    }
                    s_fx[tid] = w.x;
                    s_fy[tid] = w.y;
                    s_fz[tid] = w.z;
                    s_fp[tid] = w.txpower;
                    s_fch[tid] = w.channel;
                    fs = w.src;
                }
                s_ff[tid] = f;
                s_inv[tid] = inv;
                s_fsrc[tid] = fs;
            }
            __syncthreads();
            RM_STAMP(3); // the near frames' records are in LDS
            // from here on every wave is on its own: its links (every fourth), its pairs, their exact evaluation
            int my_np = 0;
            auto exact_mine = [&]() {
                for (int pp = lane; pp < my_np; pp += 64) {
                    const uint32_t pr = s_pairs[wave * kScPairsW + pp];
                    const int l = int(pr >> 8), c = int(pr & 0xFFu);
                    RxRecord rx_;
                    rx_.x = s_rx[l * 3 + 0];
                    rx_.y = s_rx[l * 3 + 1];
                    rx_.z = s_rx[l * 3 + 2];
                    rx_.orig = s_dst[l];
                    rx_.int_id = 0;
                    rx_.channel = s_fch[c]; // (the link was heard on the new frame's channel, and the near frames are on it)
                    rx_.enabled = 1;
                    rx_.rxprob = 1.0;
                    rm_tx_record w; // (what eval_link reads of it)
                    w.x = s_fx[c];
                    w.y = s_fy[c];
                    w.z = s_fz[c];
                    w.txpower = s_fp[c];
                    w.txprob = 1.0;
                    w.start_us = w.air_us = 0;
                    w.src = s_fsrc[c];
                    w.channel = s_fch[c];
                    const LinkEval ev = eval_link<RM_MODEL_LOGDIST, true>(m, nd, w, rx_, false);
                    if (ev.flags & kFlagInterferer) lds_add_u128(&s_acc[l * 2], q80_from_double(ev.lin));
                }
                my_np = 0;
            };
            // pair p = (link p / ns, near frame p % ns), the waves interleaved: full lanes whatever ns is
            const int n_pairs = nl * ns;
            const uint32_t inv_ns = (ns > 1) ? uint32_t((0x100000000ull + uint32_t(ns) - 1u) / uint32_t(ns)) : 0u;
            for (int p0 = wave * 64; p0 < n_pairs; p0 += 256 * kScPU) { // wave-uniform
                bool hit[kScPU];
                int pl[kScPU], pc[kScPU];
#pragma unroll
                for (int u = 0; u < kScPU; ++u) { // independent pairs: their LDS reads overlap
                    const int p = p0 + u * 256 + lane;
                    hit[u] = false;
                    pl[u] = pc[u] = 0;
                    if (p < n_pairs) {
                        const int l = (ns > 1) ? int(__umulhi(uint32_t(p), inv_ns)) : p; // p / ns (exact: p < 2^15)
                        const int c = p - l * ns;
                        pl[u] = l;
                        pc[u] = c;
                        const float4 f = s_ff[c];
                        if (f.w >= 0.f) {
                            const float4 v = s_rxf[l];
                            const int d = s_dst[l];
                            const float s2 = dist2_f32(v.x - f.x, v.y - f.y, v.z - f.z);
                            const int fsrc = s_fsrc[c];
                            hit[u] = s2 <= f.w && fsrc != d;
                            if (SHADOW && hit[u]) {
                                const int bin = min(kShadowBins - 1, int(s2 * s_inv[c]));
                                const uint32_t a = uint32_t(fsrc), b = uint32_t(d);
                                const uint64_t key = (uint64_t(a < b ? a : b) << 32) | uint64_t(a < b ? b : a);
                                hit[u] = uint32_t(mix64(m.ld_seed_mixed ^ key) >> 32) <= s_tbl[bin];
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < kScPU; ++u) {
                    const uint64_t hm = ballot64(hit[u]);
                    const int cnt = int(__popcll(hm));
                    if (cnt) {
                        if (my_np + cnt > kScPairsW) exact_mine(); // wave-uniform: room first
                        if (hit[u]) s_pairs[wave * kScPairsW + my_np + int(lane_prefix(hm))] = uint16_t((pl[u] << 8) | pc[u]);
                        my_np += cnt;
                    }
                }
            }
            RM_STAMP(4); // pairs tested
            exact_mine();
        }
        __syncthreads();
        RM_STAMP(5); // exact
    };

    for (uint32_t l0 = 0; l0 < len; l0 += kScLinks) { // block-uniform
        const int nl = int(min(uint32_t(kScLinks), len - l0));
        const bool have = tid < nl;
        const uint32_t o = src0 + l0 + uint32_t(tid);
        int node = node0, pos = pos0;
        double rssi = rssi0;
        unsigned long long self = self0;
        if (l0 != 0u) {
            __syncthreads(); // (the chunk before is done with the links in LDS)
            if (have) {
                node = t.a_dst[o];
                rssi = t.a_rssi[o];
                pos = engine_pos(nd, node);
                self = sd.self_slot[node];
            }
        }
        if (have) { // (a heard link's receiver is one of this partition's)
            double px, py, pz;
            if (nd.rec32 != nullptr) {
                const RxCompact r = nd.rec32[pos];
                px = r.x, py = r.y, pz = r.z;
            } else {
                const RxRecord r = nd.rec[pos];
                px = r.x, py = r.y, pz = r.z;
            }
            const float4 pf = nd.rxf[pos];
            // half duplex: the receiver's own frames on the air (its chain in this tick's index)
            uint32_t hd = 0u;
            if (uint32_t(self >> 32) == sd.stamp) {
                int idx = int(uint32_t(self));
                for (int hops = 0; idx >= 0 && hops <= t.n_active; ++hops) {
                    const int64_t w_start = t.tx[idx].start_us, w_end = w_start + t.tx[idx].air_us;
                    if (idx != abs_q && w_end > t.air.t_begin && w_start < q_end && w_end > wq.start_us) hd = 1u;
                    idx = sd.self_next[idx];
                }
            }
            s_dst[tid] = node;
            s_rx[tid * 3 + 0] = px;
            s_rx[tid * 3 + 1] = py;
            s_rx[tid * 3 + 2] = pz;
            s_rxf[tid] = pf;
            s_acc[tid * 2] = s_acc[tid * 2 + 1] = 0ull;
            s_hd[tid] = hd;
        }
        __syncthreads();
        RM_STAMP(2); // the links
        if (s_over == 0u) {
            int run = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                nb[w] = w * kScNearW;
                np_[w] = run;
                run += int(s_wn[w]);
            }
            np_[4] = run;
            if (run > 0) pairs_phase(run, nl);
        } else {
            // slow path (a wave's quarter overflowed: more than ~1000 co-channel frames within reach): every frame on the air
            // once more per chunk, the list filled from the front through a counter in LDS, a pair phase whenever the next
            // steps might not fit
            nb[0] = nb[1] = nb[2] = nb[3] = 0;
            if (tid == 0) s_nn = 0u;
            __syncthreads();
            for (int i0 = 0; i0 < t.n_active; i0 += 256 * kScSteps) { // block-uniform
#pragma unroll
                for (int k = 0; k < kScSteps; ++k) {
                    const int i = i0 + k * 256 + tid, ic = min(i, t.n_active - 1);
                    const bool near = i < t.n_active && is_near(i, sd.xyzr[ic], sd.ch[ic]);
                    const uint64_t hm = ballot64(near);
                    if (hm) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&s_nn, uint32_t(__popcll(hm)));
                        base = uniform_u(base);
                        if (near) s_near[base + lane_prefix(hm)] = i;
                    }
                }
                __syncthreads();
                const int nn = uniform_i(int(s_nn));
                __syncthreads(); // (everybody has looked before anybody adds again)
                if (nn + 256 * kScSteps > kScNear) { // the next steps might not fit
                    np_[0] = 0;
                    np_[1] = np_[2] = np_[3] = np_[4] = nn;
                    pairs_phase(nn, nl);
                    if (tid == 0) s_nn = 0u;
                    __syncthreads();
                }
            }
            const int nn = uniform_i(int(s_nn));
            np_[0] = 0;
            np_[1] = np_[2] = np_[3] = np_[4] = nn;
            if (nn > 0) pairs_phase(nn, nl);
        }
        __syncthreads();
        if (have) {
            U128 acc;
            acc.lo = s_acc[tid * 2];
            acc.hi = s_acc[tid * 2 + 1];
            const double sinr = rssi - 10.0 * det_log10(q80_to_double(acc) + m.ld_noise_lin);
            t.a_sinr[o] = sinr;
            if (s_hd[tid] != 0u || !(sinr >= m.ld_capture)) t.a_verdict[o] = uint8_t(RM_INTERFERED);
        }
        RM_STAMP(6);
#ifdef RM_STAMPS
        if (threadIdx.x == 0) {
            stamps[14] = __builtin_amdgcn_s_memrealtime();
            stamps[13] = uint64_t(np_[4]) | (uint64_t(len) << 32);
            stamps[12] = uint64_t(use_grid ? gw * gh : 0);
        }
#endif
    }
}

// Frames of the on-air window that had left the air when an earlier tick began are retired for good (padding records):
// rm_tick_begin's rule is applied tick by tick, and the kernels only compare with the current tick's t_begin.  Launched by
// air_tick_device when the clock goes back.
__global__ void __launch_bounds__(256) k_air_expire(rm_tx_record *recs, int n, int64_t t_seen)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && recs[i].src >= 0 && recs[i].start_us + recs[i].air_us <= t_seen) recs[i].src = -1;
}

hipError_t launch_air_expire(hipStream_t s, rm_tx_record *recs, int n, int64_t t_seen)
{
    if (n <= 0) return hipSuccess;
    RM_KLAUNCH(k_air_expire, dim3(cdiv(n, 256)), dim3(256), 0, s, recs, n, t_seen);
    return hipGetLastError();
}

hipError_t launch_sinr_scan(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const ScanDev &sd, const LaunchCfg &cfg)
{
    if (t.n_cnt <= 0) return hipSuccess;
    const dim3 grid(t.n_cnt), block(256);
    if (cfg.shadow && m.shadow_tbl) RM_KLAUNCH((k_sinr_scan<true>), grid, block, 0, s, nd, m, t, sd);
    else RM_KLAUNCH((k_sinr_scan<false>), grid, block, 0, s, nd, m, t, sd);
    return hipGetLastError();
}

} // namespace rm
