// rm_airbatch.hip -- a BATCH of SINR ticks whose frames outlive their tick (BASELINE configs[4]: 8128 us frames over 1000 us
// ticks), swept in one launch sequence (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math)
//
// The SINR extension (not reference behaviour) decides a frame's verdicts once, in its first tick,
// against the frames on the air as they stand then: the ticks of a batch depend on each other's FRAMES, never on each
// other's results.  So the heard links of all ticks come from the ordinary batch sweep of the medium without SINR
// (rm_filter.hip / rm_exact.hip / rm_reorder.hip: cut-off at the sensitivity), and the interference sums are bulk passes
// over the whole batch -- no per-receiver state, no chain of dependent round trips per tick as in the lone tick by scan
// (rm_airscan.hip), whose arithmetic this file repeats pair for pair:
//   index    every frame the batch can see (the window of frames still on the air, then the batch's ticks) gets its
//            pre-filter record at the interference level and a bin = (cell of a 64 x 64 grid over the fp32 frame, time slot),
//            slot fastest; counting sort by bin (k_ov_count, k_ov_blocksum, k_ov_scan, k_ov_fill).  The frames near a place
//            that can still be on the air in tick b are then ONE contiguous run per cell: slots slot_lo(b) .. slot(b);
//   pairs    one wave per new frame: the co-channel frames of those runs whose reach touches the circle the new frame is
//            heard in (and that overlap it in time) are its NEAR frames; (heard link, near frame) pairs pass the sweep's own
//            conservative tests (fp32 distance against the frame's cut-off, the shadowed medium's link-hash table); the
//            survivors go to a sharded list in device memory (k_ov_pairs); a frame whose pairs do not fit the list -- a dense
//            field has millions per tick -- is deferred: a second go of the same kernel evaluates its pairs where it finds them;
//   exact    one lane per surviving pair, full lanes: eval_link -- the very arithmetic of the list and scan forms -- and the
//            linear power added in Q80 fixed point to the link's 128-bit sum (two 64-bit atomics with the carry counted:
//            exact whatever the order) (k_ov_exact);
//   verdict  sinr = rssi - 10 log10(sum + noise), capture threshold, half duplex (k_ov_verdict).
// A new frame looks only at frames of its own and earlier slots (causal), and every frame's own times are tested: the
// slot range only has to be conservative.
#include "rm_device.hpp"

#include <stdlib.h>

namespace rm {

constexpr int kOvW = 4;          // new frames per workgroup: one per wave
constexpr int kOvLinks = 64;     // heard links of the frame handled together (one per lane)
constexpr int kOvNear = 128;     // near frames gathered between two pair phases
constexpr int kOvStage = 128;    // surviving pairs a wave holds before it writes them out
#ifndef RM_OV_PU
#define RM_OV_PU 1
#endif
#ifndef RM_OV_OCC
#define RM_OV_OCC 6
#endif
constexpr int kOvPU = RM_OV_PU;  // pairs per lane tested together (2: 657 instead of 632 us per 128 ticks of configs[4] -- spills at 80 VGPRs)
constexpr int kOvOcc = RM_OV_OCC; // workgroups of k_ov_pairs per CU the register budget is set for (with a resident grid of as many: 4 / 5 / 6 / 7 -> 724 / 649 / 627 / 707 us per 128 ticks of configs[4]; 7 spills 25 VGPRs)
static_assert(kOvNear >= 2 * 64, "room for 64 more candidates whenever the list holds at most kOvNear - 64");
static_assert(kOvStage >= 2 * 64, "room for 64 more pairs whenever the stage holds at most kOvStage - 64");
static_assert(kOvLinks * kOvNear <= (1 << 16), "pair indices are divided by multiplication");

RM_D int ov_cell1(float x, float half, float inv)
{
    const int c = int((x + half) * inv);
    return min(max(c, 0), kSgG - 1);
}

// ---- the batch's descriptors --------------------------------------------------------------------------------------------
// One launch in front of the batch: the ticks' descriptors and the slots' first frames come out of the host's pinned block (the
// device reads it itself, as k_fetch_ticks does), the batch's counters start at zero.  (Two copies and two fills on the stream
// used to do this, and a third copy took the overflow flag back: five stream operations between two batches, 50 us of an idle
// device each time; the flag now goes to the host's word from k_ov_verdict.)
__global__ void __launch_bounds__(256) k_ov_begin(const OvTick *__restrict__ h_ticks, const int32_t *__restrict__ h_first, int n_ticks, int n_slots,
                                                  OvTick *ticks, int32_t *slot_first, uint32_t *misc, uint32_t *pair_tail)
{
    const int tid = int(blockIdx.x) * 256 + int(threadIdx.x), step = int(gridDim.x) * 256;
    static_assert(sizeof(OvTick) % 4 == 0, "");
    const uint32_t *src = reinterpret_cast<const uint32_t *>(h_ticks);
    uint32_t *dst = reinterpret_cast<uint32_t *>(ticks);
    for (int i = tid; i < n_ticks * int(sizeof(OvTick) / 4); i += step) dst[i] = src[i];
    for (int i = tid; i <= n_slots; i += step) slot_first[i] = h_first[i];
    for (int i = tid; i < 8 + kSgMax; i += step) misc[i] = 0u;
    for (int i = tid; i < kShards * kShardStride; i += step) pair_tail[i] = 0u;
}

hipError_t launch_ov_begin(hipStream_t s, const OvTick *h_ticks, const int32_t *h_first, int n_ticks, int n_slots, OvTick *ticks,
                           int32_t *slot_first, uint32_t *misc, uint32_t *pair_tail)
{
    RM_KLAUNCH(k_ov_begin, dim3(max(1, min(64, cdiv(n_ticks * int(sizeof(OvTick) / 4), 256)))), dim3(256), 0, s, h_ticks, h_first, n_ticks,
               n_slots, ticks, slot_first, misc, pair_tail);
    return hipGetLastError();
}

// ---- index ----------------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_ov_count(const NodesDev nd, const ModelDev m, const OvDev ov)
{
    const int slot = blockIdx.y;
    const int f0 = ov.slot_first[slot], f1 = ov.slot_first[slot + 1];
    const int i = f0 + int(blockIdx.x) * 256 + int(threadIdx.x);
    float radius = 0.f; // of a frame that goes into the grid (its reach at the interference level)
    // (a tick swept over a rank's frame list, k_rank_frames: its slot holds n_new frames -- the device's count -- and padding behind them)
    const int bt = slot - (ov.n_slots - ov.n_ticks);
    const int live = (bt >= 0) ? min(f1 - f0, ov.ticks[bt].n_new) : f1 - f0;
    if (bt >= 0) {
        // the pair stage's items: this tick's frames behind those of the ticks before it (their counts summed here: at most
        // RM_MAX_BATCH words), tick << 16 | frame; the last tick's first workgroup publishes the total
        __shared__ uint32_t s_part[4];
        uint32_t before = 0;
        for (int b = int(threadIdx.x); b < bt; b += 256) before += uint32_t(max(ov.ticks[b].n_new, 0));
        for (int d = 32; d >= 1; d >>= 1) before += uint32_t(__shfl_xor(int(before), d));
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = before;
        __syncthreads();
        before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        const int li = int(blockIdx.x) * 256 + int(threadIdx.x);
        if (li < live) ov.items[before + uint32_t(li)] = (uint32_t(bt) << 16) | uint32_t(li);
        if (bt == ov.n_ticks - 1 && blockIdx.x == 0 && threadIdx.x == 0) ov.misc[3] = before + uint32_t(max(live, 0));
    }
    if (i < f1) { // (no early return: the wave reduces the radii together below)
        rm_tx_record r{};
        r.src = -1;
        if (i - f0 < live) r = ov.tx[i];
        float4 f;
        double thr64;
        tx_prefilter(m, r, f, thr64);
        const bool valid = r.src >= 0 && r.src < nd.n;
        if (!valid) f.w = -1.f;
        float inv = 0.f;
        if (m.shadow_tbl && f.w > 0.f && f.w < __builtin_inff()) { // (the sweep's second-level filter applies to this frame)
            const float cut = __builtin_sqrtf(f.w);
            if (1.01f * (2.0f * float(m.f32_slack)) / (0.15f * cut) + 1e-5f <= float(kShadowPad)) inv = float(kShadowBins) / f.w;
        }
        ov.fr_f[i] = f;
        ov.fr_m[i] = make_int4(__float_as_int(inv), r.src, r.channel, i);
        ov.fr_t[i] = make_longlong2(r.start_us, r.start_us + r.air_us);
        uint32_t bin = 0xFFFFFFFFu;
        int next = -1;
        if (valid) {
            // the node's chain of frames (half duplex does not ask for reach)
            const unsigned long long stamp = (unsigned long long)ov.stamp << 32;
            const unsigned long long old = atomicExch(&ov.self_slot[r.src], stamp | (unsigned long long)uint32_t(i));
            if ((old >> 32) == ov.stamp) next = int(uint32_t(old));
            if (f.w >= 0.f) {
                if (f.w < __builtin_inff()) {
                    const int cell = ov_cell1(f.y, ov.half, ov.inv) * kSgG + ov_cell1(f.x, ov.half, ov.inv);
                    bin = uint32_t(cell) * uint32_t(ov.n_slots) + uint32_t(slot);
                    atomicAdd(&ov.bin_cnt[bin], 1u);
                    radius = sqrt_up(f.w);
                } else {
                    ov.every[atomicAdd(&ov.misc[0], 1u)] = uint32_t(i); // no bound, or outside the fp32 frame: everybody looks at it
                }
            }
        }
        ov.self_next[i] = next;
        ov.fr_bin[i] = bin;
        ov.defer[i] = 0;
    }
    // the largest radius on the air: one atomic per WAVE (radii are >= 0: their bits order like they do) -- one per frame on 64
    // words was 8000 atomics per word and batch, most of this kernel's time
    radius = wave_max(radius);
    if ((threadIdx.x & 63) == 0 && radius > 0.f) atomicMax(&ov.misc[8 + ((blockIdx.x * 4 + (threadIdx.x >> 6) + blockIdx.y * 7) & (kSgMax - 1))], __float_as_uint(radius));
}

// sum of one block of kOvScanBlock bins
__global__ void __launch_bounds__(256) k_ov_blocksum(const OvDev ov)
{
    __shared__ uint32_t s_w[4];
    const int base = int(blockIdx.x) * kOvScanBlock;
    uint32_t sum = 0;
    for (int k = int(threadIdx.x); k < kOvScanBlock; k += 256) {
        const int i = base + k;
        if (i < ov.n_bins) sum += ov.bin_cnt[i];
    }
    for (int d = 32; d >= 1; d >>= 1) sum += uint32_t(__shfl_xor(int(sum), d));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) ov.block_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of the bins' counts: the block's base from the block sums before it, then its own kOvScanBlock bins
__global__ void __launch_bounds__(256) k_ov_scan(const OvDev ov)
{
    __shared__ uint32_t s_w[4], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t part = 0;
    for (int j = tid; j < int(blockIdx.x); j += 256) part += ov.block_sum[j];
    for (int d = 32; d >= 1; d >>= 1) part += uint32_t(__shfl_xor(int(part), d));
    if (lane == 0) s_w[wave] = part;
    __syncthreads();
    if (tid == 0) s_base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    constexpr int kPer = kOvScanBlock / 256;
    const int i0 = int(blockIdx.x) * kOvScanBlock + tid * kPer;
    uint32_t v[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        v[k] = (i0 + k < ov.n_bins) ? ov.bin_cnt[i0 + k] : 0u;
        sum += v[k];
    }
    const uint32_t inc = wave_inclusive_scan(sum, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t run = s_base + inc - sum;
    for (int w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        if (i0 + k < ov.n_bins) ov.bin_off[i0 + k] = run;
        run += v[k];
    }
    if (i0 < ov.n_bins && i0 + kPer >= ov.n_bins) ov.bin_off[ov.n_bins] = run; // (the thread that holds the last bin)
}

// every binned frame takes a place in its bin's run (the counts go back to zero: nothing to clear for the next batch)
__global__ void __launch_bounds__(256) k_ov_fill(const OvDev ov)
{
    const int i = int(blockIdx.x) * 256 + int(threadIdx.x);
    if (i >= ov.n_frames) return;
    const uint32_t bin = ov.fr_bin[i];
    if (bin == 0xFFFFFFFFu) return;
    const uint32_t k = atomicSub(&ov.bin_cnt[bin], 1u) - 1u;
    const uint32_t p = ov.bin_off[bin] + k;
    ov.e_f[p] = ov.fr_f[i];
    ov.e_m[p] = ov.fr_m[i];
    ov.e_t[p] = ov.fr_t[i];
}

// ---- pairs ----------------------------------------------------------------------------------------------------------
// One WAVE per new frame (four frames per workgroup, each wave on its own: no workgroup barrier after the first).  What a
// frame costs is a chain of dependent round trips -- its links' receivers (node -> engine position -> pre-filter record,
// node -> chain of own frames) and its near frames (cells' runs -> entries) -- and the chip hides such chains only behind
// other frames' chains: a 256-thread workgroup per frame kept 6 frames per CU in flight, a wave per frame keeps 30.

// fence between a wave's LDS writes and its own lanes' reads of them (the LDS executes a wave's accesses in order; this
// keeps the compiler from moving them across)
RM_D void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// INLINE: the second go for the frames whose pairs did not all fit the list (OvDev::defer) -- their links' sums are formed
// from scratch, every surviving pair evaluated where it is found (what k_ov_exact has added for such a frame is discarded).
template <bool SHADOW, bool INLINE>
__global__ void __launch_bounds__(256, INLINE ? 2 : kOvOcc) k_ov_pairs(const NodesDev nd, const ModelDev m, const OvDev ov)
{
    __shared__ float4 s_rxf[kOvW][kOvLinks];   // the links' receivers in the fp32 frame; .w: the node index's bits (one read per pair)
    __shared__ int s_pos[kOvW][kOvLinks];      // ... engine position
    __shared__ float4 s_ff[kOvW][kOvNear];     // the near frames: pre-filter record at the interference level
    __shared__ float s_inv[kOvW][kOvNear];
    __shared__ int s_src[kOvW][kOvNear], s_idx[kOvW][kOvNear];
    __shared__ uint32_t s_stage[kOvW][kOvStage]; // surviving pairs: link << 16 | near frame
    __shared__ uint32_t s_tbl[SHADOW ? kShadowBins : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    if (INLINE && ov.misc[1] == 0u) return;      // nothing was deferred (the whole grid leaves at once)
    if (SHADOW) s_tbl[tid] = m.shadow_tbl[tid]; // kBlock == kShadowBins
    __syncthreads();                             // (the only one: from here on every wave is on its own)
    // A resident grid: a wave takes every (gridDim.x * kOvW)-th (tick, frame) item.  (One workgroup per four frames was 16 000
    // workgroups per launch of 64 ticks, and the launch took as long as the same grid of workgroups that leave at once:
    // it was bound by the rate at which workgroups with 23 KB of LDS can be placed, not by what they did.)
    // The (tick, frame) items, dense and tick-major (OvDev::items, written by k_ov_count: a tick's frame count is the device's -- a
    // rank's frame list keeps a fraction of the gathered slots).  A wave takes every (gridDim.x * kOvW)-th item, so at any moment
    // the whole chip works on a few neighbouring ticks and their part of the index stays in the L2.  (Dealing the waves to the
    // ticks instead -- every tick of the batch in flight at once -- was measured: 792 against 630 us per 128 ticks.)
    const int items = uniform_i(int(ov.misc[3]));
    for (int item = int(blockIdx.x) * kOvW + wave; item < items; item += int(gridDim.x) * kOvW) { // wave-uniform
    const uint32_t it = uniform_u(ov.items[item]);
    const int b_tick = int(it >> 16), q = int(it & 0xFFFFu);
    const OvTick &tk = ov.ticks[b_tick];
    if (INLINE && ov.defer[tk.frame_first + q] == 0) continue;
    const uint32_t l_first = uniform_u(tk.slot_off[tk.shift + q]);
    const uint32_t len = uniform_u(tk.slot_off[tk.shift + q + 1]) - l_first;
    if (len == 0u || tk.flags[1] != 0u) continue; // (a tick dropped for capacity has no links to speak of)
    const int abs_q = tk.frame_first + q;
    const int vis_end = tk.frame_first + tk.n_new; // the frames of its own and earlier slots
    const rm_tx_record wq = ov.tx[abs_q];
    const int64_t q_end = wq.start_us + wq.air_us, t_begin = tk.t_begin;
    const int n_slots = ov.n_slots, slot_lo = tk.slot_lo, slot_hi = tk.slot;
    float rmax = __uint_as_float(ov.misc[8 + lane]); // kSgMax == 64: one slot per lane
    rmax = wave_max(rmax);
    const int first_lo = ov.slot_first[slot_lo];
    const int n_every = uniform_i(int(ov.misc[0]));

    // the circle this frame is heard in: its cut-off at the sensitivity level, as the sweep used it
    float4 fq;
    {
        double thr64;
        tx_prefilter_at(m, m.ld_sens, wq, fq, thr64);
    }
    const float rq = (fq.w >= 0.f) ? ((fq.w < __builtin_inff()) ? sqrt_up(fq.w) : __builtin_inff()) : 0.f;
    // near: co-channel, on the air and overlapping the new frame in time (air_sinr's test), visible (its own or an earlier
    // slot), and its reach touches that circle (both radii carry the fp32 frame's slack; the margin covers this test's rounding)
    auto is_near = [&](const float4 &f, const int4 &mm, const longlong2 &tt) -> bool {
        if (!(mm.w != abs_q && mm.w < vis_end && f.w >= 0.f && mm.z == wq.channel)) return false;
        if (!(tt.y > t_begin && tt.x < q_end && tt.y > wq.start_us)) return false;
        if (!(f.w < __builtin_inff())) return true;
        const float reach = rq + sqrt_up(f.w);
        return dist2_f32(f.x - fq.x, f.y - fq.y, f.z - fq.z) <= reach * reach * (1.0f + 1e-5f);
    };
    // which cells can hold a near frame?  |dx| <= rq + (largest radius), and positions map to cells monotonically
    bool use_grid = rq < __builtin_inff() && rmax < __builtin_inff();
    int cx0 = 0, cy0 = 0, gw = 1, gh = 1;
    if (use_grid) {
        const float reach = (rq + rmax) * (1.0f + 2e-5f) + 1e-3f / ov.inv;
        cx0 = ov_cell1(fq.x - reach, ov.half, ov.inv);
        cy0 = ov_cell1(fq.y - reach, ov.half, ov.inv);
        gw = ov_cell1(fq.x + reach, ov.half, ov.inv) - cx0 + 1;
        gh = ov_cell1(fq.y + reach, ov.half, ov.inv) - cy0 + 1;
    }
    use_grid = use_grid && gw * gh * 2 <= (vis_end - first_lo) + 64; // (otherwise looking at every visible frame is less work)
    use_grid = __builtin_amdgcn_readfirstlane(int(use_grid)) != 0;
    const int n_cells = uniform_i(gw * gh);

    // ---- the wave's staged pairs go to the shards' regions of the pair list: one atomic per piece; a full shard hands the
    // rest to the next one (the list as a whole is sized for the batch; only when every shard is full does the batch fail)
    int my_ns = 0;          // staged pairs (wave-uniform)
    uint32_t l_base = 0;    // first link of the chunk in the tick's compact arrays
    uint32_t shard = (uint32_t(item) * 2654435761u >> 20) & uint32_t(kShards - 1);
    auto flush = [&]() {
        if (INLINE) { // the staged pairs are evaluated here, full lanes: eval_link, the sum in Q80 (k_ov_exact's arithmetic)
            for (int k = lane; k < my_ns; k += 64) {
                const uint32_t pr = s_stage[wave][k];
                const int l = int(pr >> 16), c = int(pr & 0xFFFFu);
                const int pos = s_pos[wave][l];
                RxRecord rx_;
                if (nd.rec32 != nullptr) {
                    const RxCompact r = nd.rec32[pos];
                    rx_.x = r.x, rx_.y = r.y, rx_.z = r.z;
                    rx_.orig = r.orig;
                } else {
                    const RxRecord r = nd.rec[pos];
                    rx_.x = r.x, rx_.y = r.y, rx_.z = r.z;
                    rx_.orig = r.orig;
                }
                const rm_tx_record w = ov.tx[s_idx[wave][c]];
                rx_.int_id = 0;
                rx_.channel = w.channel;
                rx_.enabled = 1;
                rx_.rxprob = 1.0;
                const LinkEval ev = eval_link<RM_MODEL_LOGDIST, true>(m, nd, w, rx_, false);
                if (ev.flags & kFlagInterferer) {
                    const U128 v = q80_from_double(ev.lin);
                    if ((v.lo | v.hi) != 0ull) {
                        const uint32_t o = l_base + uint32_t(l);
                        const unsigned long long old = atomicAdd(&tk.acc_lo[o], (unsigned long long)v.lo);
                        const unsigned long long carry = (old + v.lo < old) ? 1ull : 0ull;
                        if (v.hi + carry) atomicAdd(&tk.acc_hi[o], (unsigned long long)(v.hi + carry));
                    }
                }
            }
            my_ns = 0;
            return;
        }
        int done = 0;
        for (int tries = 0; done < my_ns && tries < kShards; ++tries) { // wave-uniform
            const uint32_t want = uint32_t(my_ns - done);
            uint32_t base = 0;
            if (lane == 0) {
                // (a shard that is known to be full is not asked again: its counter would only run away)
                base = ov.pair_tail[shard * kShardStride] >= ov.pair_seg ? ov.pair_seg : atomicAdd(&ov.pair_tail[shard * kShardStride], want);
            }
            base = uniform_u(base);
            const uint32_t fit = base < ov.pair_seg ? min(want, ov.pair_seg - base) : 0u;
            for (uint32_t k = uint32_t(lane); k < fit; k += 64u) {
                const uint32_t pr = s_stage[wave][uint32_t(done) + k];
                const int l = int(pr >> 16), c = int(pr & 0xFFFFu);
                OvPair p;
                p.link = l_base + uint32_t(l);
                p.tick = uint32_t(b_tick);
                p.pos = s_pos[wave][l];
                p.frame = s_idx[wave][c];
                ov.pairs[size_t(shard) * ov.pair_seg + base + k] = p;
            }
            done += int(fit);
            if (done < my_ns) shard = (shard + 1u) & uint32_t(kShards - 1);
        }
        if (done < my_ns && lane == 0) { // the whole list is full: this frame's sums are formed by the second go (k_ov_pairs<., true>)
            ov.defer[abs_q] = 1;
            ov.misc[1] = 1u;
        }
        my_ns = 0;
    };

    int n_near = 0; // near frames in the list (wave-uniform)
    int nl = 0;     // links of the chunk in LDS
    // the near frames gathered so far against the chunk's links.  (The link-hash test of the shadowed medium -- two 64-bit
    // multiplies per lane -- is paid by every wave iteration in which some lane passed the distance test; taking the
    // distance hits aside through LDS and hashing them with full lanes was built and measured: 109 instead of 75 VGPRs,
    // 4 instead of 6 waves per SIMD, 502 instead of 328 us per 64 ticks.  Not in the tree.)
    auto pairs_phase = [&]() {
        wave_lds_fence();
        const int ns = n_near;
        if (ns > 0 && nl > 0) {
            const int n_pairs = nl * ns;
            const uint32_t inv_ns = (ns > 1) ? uint32_t((0x100000000ull + uint32_t(ns) - 1u) / uint32_t(ns)) : 0u;
            for (int p0 = 0; p0 < n_pairs; p0 += 64 * kOvPU) { // wave-uniform
                bool hit[kOvPU];
                int pl[kOvPU], pc[kOvPU];
#pragma unroll
                for (int u = 0; u < kOvPU; ++u) { // independent pairs: their LDS reads overlap
                    const int p = p0 + u * 64 + lane;
                    hit[u] = false;
                    pl[u] = pc[u] = 0;
                    if (p < n_pairs) {
                        const int l = (ns > 1) ? int(__umulhi(uint32_t(p), inv_ns)) : p; // p / ns (exact: p < 2^16)
                        const int c = p - l * ns;
                        pl[u] = l;
                        pc[u] = c;
                        const float4 f = s_ff[wave][c];
                        const float4 v = s_rxf[wave][l];
                        const int d = __float_as_int(v.w);
                        const float s2 = dist2_f32(v.x - f.x, v.y - f.y, v.z - f.z);
                        const int fsrc = s_src[wave][c];
                        hit[u] = s2 <= f.w && fsrc != d;
                        if (SHADOW && hit[u]) {
                            const int bin = min(kShadowBins - 1, int(s2 * s_inv[wave][c]));
                            const uint32_t a = uint32_t(fsrc), b = uint32_t(d);
                            const uint64_t key = (uint64_t(a < b ? a : b) << 32) | uint64_t(a < b ? b : a);
                            hit[u] = uint32_t(mix64(m.ld_seed_mixed ^ key) >> 32) <= s_tbl[bin];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < kOvPU; ++u) {
                    const uint64_t hm = ballot64(hit[u]);
                    const int cnt = int(__popcll(hm));
                    if (cnt) {
                        if (my_ns + cnt > kOvStage) { // wave-uniform: room first
                            wave_lds_fence();
                            flush();
                        }
                        if (hit[u]) s_stage[wave][my_ns + int(lane_prefix(hm))] = (uint32_t(pl[u]) << 16) | uint32_t(pc[u]);
                        my_ns += cnt;
                    }
                }
            }
            wave_lds_fence();
            flush(); // (the staged pairs name near frames by their place in the list that is replaced next)
        }
        n_near = 0;
    };
    // one candidate frame per lane: the near ones join the list (emptied by a pair phase when the next 64 might not fit)
    auto consider = [&](const bool have, const float4 &f, const int4 &mm, const longlong2 &tt) {
        if (n_near + 64 > kOvNear) pairs_phase();
        const bool near = have && is_near(f, mm, tt);
        const uint64_t hm = ballot64(near);
        if (near) {
            const int k = n_near + int(lane_prefix(hm));
            s_ff[wave][k] = f;
            s_inv[wave][k] = __int_as_float(mm.x);
            s_src[wave][k] = mm.y;
            s_idx[wave][k] = mm.w;
        }
        n_near += int(__popcll(hm));
    };

    for (uint32_t l0 = 0; l0 < len; l0 += kOvLinks) { // wave-uniform
        if (!INLINE && l0 != 0u && uniform_i(int(ov.defer[abs_q])) != 0) break; // (deferred: the second go does all of this frame)
        nl = int(min(uint32_t(kOvLinks), len - l0));
        l_base = l_first + l0;
        wave_lds_fence(); // (the chunk before is done with the links in LDS)
        // the links' first round trip; the near frames' first one goes out with it (the chunk's first 64 cells)
        const uint32_t o = l_base + uint32_t(lane);
        const bool have_l = lane < nl;
        int node = 0;
        if (have_l) node = tk.out_dst[o];
        uint32_t c_lo = 0, c_cnt = 0;
        const uint32_t inv_gw = (gw > 1) ? uint32_t((0x100000000ull + uint32_t(gw) - 1u) / uint32_t(gw)) : 0u;
        auto cell_run = [&](const int cl_i, uint32_t &lo, uint32_t &cnt) {
            lo = cnt = 0u;
            if (cl_i < n_cells) {
                const uint32_t cl = uint32_t(cl_i);
                const uint32_t cyl = (gw > 1) ? __umulhi(cl, inv_gw) : cl; // cl / gw (exact: cl < 4096)
                const uint32_t cell = (uint32_t(cy0) + cyl) * uint32_t(kSgG) + uint32_t(cx0) + (cl - cyl * uint32_t(gw));
                lo = ov.bin_off[cell * uint32_t(n_slots) + uint32_t(slot_lo)];
                cnt = ov.bin_off[cell * uint32_t(n_slots) + uint32_t(slot_hi) + 1u] - lo; // (slot fastest: the run of slots is contiguous)
            }
        };
        if (use_grid) cell_run(lane, c_lo, c_cnt);
        // second round trip of the links: engine position, the receiver's chain of own frames
        int pos = -1;
        unsigned long long self = 0ull;
        if (have_l) {
            pos = engine_pos(nd, node); // (a heard link's receiver is one of this partition's)
            self = ov.self_slot[node];
        }
        float4 rxf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have_l) rxf = nd.rxf[pos];
        if (have_l) {
            // half duplex: the receiver's own frames on the air (its chain in this batch's index)
            uint8_t hd = 0;
            if (uint32_t(self >> 32) == ov.stamp) {
                int idx = int(uint32_t(self));
                for (int hops = 0; idx >= 0 && hops <= ov.n_frames; ++hops) {
                    if (idx != abs_q && idx < vis_end) {
                        const longlong2 tt = ov.fr_t[idx];
                        if (tt.y > t_begin && tt.x < q_end && tt.y > wq.start_us) hd = 1;
                    }
                    idx = ov.self_next[idx];
                }
            }
            tk.hd[o] = hd;
            tk.acc_lo[o] = 0ull;
            tk.acc_hi[o] = 0ull;
            s_pos[wave][lane] = pos;
            rxf.w = __int_as_float(node); // (the channel bits are not looked at here: the link was heard on the frame's channel)
            s_rxf[wave][lane] = rxf;
        }
        n_near = 0;
        if (use_grid) {
            for (int c0 = 0; c0 < n_cells; c0 += 64) { // wave-uniform; a chunk of 64 cells, one per lane
                if (c0 > 0) cell_run(c0 + lane, c_lo, c_cnt);
                const uint32_t inc = wave_inclusive_scan(c_cnt, lane);
                const uint32_t excl = inc - c_cnt;
                const uint32_t total = uniform_u(uint32_t(__shfl(int(inc), 63)));
                for (uint32_t j0 = 0; j0 < total; j0 += 64u) { // wave-uniform
                    const uint32_t j = j0 + uint32_t(lane);
                    const bool have = j < total;
                    // the cell whose run holds entry j: the last lane with excl <= j (the prefix is non-decreasing over the lanes)
                    int a = 0;
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1) {
                        const int cand = a + step;
                        const uint32_t e = uint32_t(__shfl(int(excl), min(cand, 63)));
                        if (cand < 64 && e <= j) a = cand;
                    }
                    const uint32_t lo_a = uint32_t(__shfl(int(c_lo), a)), ex_a = uint32_t(__shfl(int(excl), a));
                    float4 f = make_float4(0.f, 0.f, 0.f, -1.f);
                    int4 mm = make_int4(0, -1, 0, abs_q);
                    longlong2 tt = make_longlong2(0, 0);
                    if (have) {
                        const uint32_t p = lo_a + (j - ex_a);
                        f = ov.e_f[p];
                        mm = ov.e_m[p];
                        tt = ov.e_t[p];
                    }
                    consider(have, f, mm, tt);
                }
            }
            for (int e0 = 0; e0 < n_every; e0 += 64) { // frames without a cell
                const bool have = e0 + lane < n_every;
                const int i = have ? int(ov.every[e0 + lane]) : abs_q;
                consider(have, ov.fr_f[i], ov.fr_m[i], ov.fr_t[i]);
            }
        } else {
            for (int i0 = first_lo; i0 < vis_end; i0 += 64) { // wave-uniform
                const bool have = i0 + lane < vis_end;
                const int i = have ? i0 + lane : abs_q;
                consider(have, ov.fr_f[i], ov.fr_m[i], ov.fr_t[i]);
            }
        }
        pairs_phase();
    }
    } // (items)
}

// ---- exact ----------------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_ov_exact(const NodesDev nd, const ModelDev m, const OvDev ov)
{
    const uint32_t sh = blockIdx.y;
    const uint32_t n = min(ov.pair_tail[sh * kShardStride], ov.pair_seg);
    const int lane = threadIdx.x & 63;
    uint32_t n_int = 0; // pairs of this thread that did interfere (statistics: OvDev::misc[2])
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) { // (whole waves stay in the loop: they reduce together)
        const uint32_t i = i0 + threadIdx.x;
        const bool have = i < n;
        OvPair p;
        p.link = 0xFFFFFFFFu, p.tick = 0xFFFFFFFFu, p.pos = 0, p.frame = 0;
        U128 v = {0, 0};
        if (have) {
            p = ov.pairs[size_t(sh) * ov.pair_seg + i];
            RxRecord rx_;
            if (nd.rec32 != nullptr) {
                const RxCompact r = nd.rec32[p.pos];
                rx_.x = r.x, rx_.y = r.y, rx_.z = r.z;
                rx_.orig = r.orig;
            } else {
                const RxRecord r = nd.rec[p.pos];
                rx_.x = r.x, rx_.y = r.y, rx_.z = r.z;
                rx_.orig = r.orig;
            }
            const rm_tx_record w = ov.tx[p.frame];
            rx_.int_id = 0;
            rx_.channel = w.channel; // (the link was heard on the new frame's channel, and its near frames are on it)
            rx_.enabled = 1;
            rx_.rxprob = 1.0;
            const LinkEval ev = eval_link<RM_MODEL_LOGDIST, true>(m, nd, w, rx_, false);
            if (ev.flags & kFlagInterferer) {
                v = q80_from_double(ev.lin);
                ++n_int;
            }
        }
        // a link's pairs are neighbours in the list (a frame's wave stages them link by link): the wave adds up every run of
        // equal (tick, link) first, and the run's last lane adds the run's sum to the link's accumulator -- two 64-bit atomics
        // per run instead of per pair; Q80 sums are exact, so the grouping changes nothing
        const uint32_t pl = uint32_t(__shfl_up(int(p.link), 1)), pt = uint32_t(__shfl_up(int(p.tick), 1));
        const uint64_t starts = ballot64(lane == 0 || pl != p.link || pt != p.tick);
        const uint64_t upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        const int run_start = 63 - __clzll((long long)(starts & upto));
        const bool run_last = lane == 63 || ((starts >> (lane + 1)) & 1ull) != 0ull;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            U128 o;
            o.lo = (uint64_t(uint32_t(__shfl_up(int(uint32_t(v.lo >> 32)), d))) << 32) | uint64_t(uint32_t(__shfl_up(int(uint32_t(v.lo)), d)));
            o.hi = (uint64_t(uint32_t(__shfl_up(int(uint32_t(v.hi >> 32)), d))) << 32) | uint64_t(uint32_t(__shfl_up(int(uint32_t(v.hi)), d)));
            if (lane - d >= run_start) v = u128_add(v, o);
        }
        if (have && run_last && (v.lo | v.hi) != 0ull) {
            const OvTick &tk = ov.ticks[p.tick];
            const unsigned long long old = atomicAdd(&tk.acc_lo[p.link], (unsigned long long)v.lo);
            const unsigned long long carry = (old + v.lo < old) ? 1ull : 0ull; // (the low words' running sum is exact mod 2^64: so is the carry count)
            if (v.hi + carry) atomicAdd(&tk.acc_hi[p.link], (unsigned long long)(v.hi + carry));
        }
    }
    // statistics: how many of the evaluated pairs were interferers (one atomic per workgroup)
    __shared__ uint32_t s_int[4];
    for (int d = 32; d >= 1; d >>= 1) n_int += uint32_t(__shfl_xor(int(n_int), d));
    if (lane == 0) s_int[threadIdx.x >> 6] = n_int;
    __syncthreads();
    if (threadIdx.x == 0 && (s_int[0] | s_int[1] | s_int[2] | s_int[3])) atomicAdd(&ov.misc[2], s_int[0] + s_int[1] + s_int[2] + s_int[3]);
}

// ---- verdicts -------------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_ov_verdict(const ModelDev m, const OvDev ov)
{
    // (frames were deferred -- the pair list was full: the host grows it for a later batch.  Its word, in pinned memory.)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && ov.h_flag != nullptr)
        *reinterpret_cast<volatile uint32_t *>(ov.h_flag) = ov.misc[1]; // (read two batches later, behind an event: no fence, no write-back of the L2 for it)
    const OvTick &tk = ov.ticks[blockIdx.y];
    if (tk.n_new <= 0) return;
    if (tk.flags[1] != 0u) return;
    const uint32_t total = tk.slot_off[tk.shift + tk.n_new];
    for (uint32_t o = blockIdx.x * 256u + threadIdx.x; o < total; o += gridDim.x * 256u) {
        U128 acc;
        acc.lo = tk.acc_lo[o];
        acc.hi = tk.acc_hi[o];
        const double sinr = tk.out_rssi[o] - 10.0 * det_log10(q80_to_double(acc) + m.ld_noise_lin);
        tk.out_sinr[o] = sinr;
        if (tk.hd[o] != 0 || !(sinr >= m.ld_capture)) tk.out_verdict[o] = uint8_t(RM_INTERFERED);
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------

hipError_t launch_ov_index(hipStream_t s, const NodesDev &nd, const ModelDev &m, const OvDev &ov, int max_slot_frames)
{
    if (ov.n_frames <= 0 || ov.n_slots <= 0) return hipSuccess;
    const int nb = cdiv(ov.n_bins, kOvScanBlock);
    RM_KLAUNCH(k_ov_count, dim3(cdiv(max(max_slot_frames, 1), 256), ov.n_slots), dim3(256), 0, s, nd, m, ov);
    RM_KLAUNCH(k_ov_blocksum, dim3(nb), dim3(256), 0, s, ov);
    RM_KLAUNCH(k_ov_scan, dim3(nb), dim3(256), 0, s, ov);
    RM_KLAUNCH(k_ov_fill, dim3(cdiv(ov.n_frames, 256)), dim3(256), 0, s, ov);
    return hipGetLastError();
}

hipError_t launch_ov_sinr(hipStream_t s, const NodesDev &nd, const ModelDev &m, const OvDev &ov, int max_new, int max_links, const LaunchCfg &cfg)
{
    if (ov.n_ticks <= 0 || max_new <= 0) return hipSuccess;
    // a resident grid: as many workgroups as the chip holds at once (six per CU), each wave striding over the (tick, frame) items
    int wgs = 256 * 6;
    if (const char *e = getenv("RM_OV_PAIRS_WGS")) wgs = max(1, atoi(e));
    const dim3 grid(max(1, min(wgs, cdiv(max_new * ov.n_ticks, kOvW)))), block(256);
    const bool sh = cfg.shadow && m.shadow_tbl;
    if (sh) RM_KLAUNCH((k_ov_pairs<true, false>), grid, block, 0, s, nd, m, ov);
    else RM_KLAUNCH((k_ov_pairs<false, false>), grid, block, 0, s, nd, m, ov);
    int gx = 16; // (workgroups per shard; measured on configs[4], 128 ticks: 4 / 8 / 16 / 32 / 64 -> 475 / 440 / 389 / 387 / 386 us; a rank's share at 512: 2 / 8 / 16 -> 366 / 225 / 198)
    if (const char *e = getenv("RM_OV_EXACT_GX")) gx = max(1, atoi(e));
    RM_KLAUNCH(k_ov_exact, dim3(gx, kShards), dim3(256), 0, s, nd, m, ov);
    // the frames whose pairs did not fit the list (none, normally: the grid leaves at once -- one workgroup per CU, so that
    // placing it costs little; the 1536 of the first go took 35 us to come and go)
    const dim3 grid2(min(int(grid.x), 256));
    if (sh) RM_KLAUNCH((k_ov_pairs<true, true>), grid2, block, 0, s, nd, m, ov);
    else RM_KLAUNCH((k_ov_pairs<false, true>), grid2, block, 0, s, nd, m, ov);
    // (a receiver partition hears 1 / share of a tick's links: as many workgroups per tick would each find a handful)
    const int share = (nd.n_rx > 0 && nd.n_rx < nd.n) ? max(1, nd.n / nd.n_rx) : 1;
    RM_KLAUNCH(k_ov_verdict, dim3(max(1, min(max(4, 64 / share), cdiv(max(max_links, 1), 256))), ov.n_ticks), dim3(256), 0, s, m, ov);
    return hipGetLastError();
}

} // namespace rm
