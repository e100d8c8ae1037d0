// rm_transmit.hip -- the per-packet call (k_transmit_one), batch descriptors, host exports of the exact math
// (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math; overview at the top of rm_engine.h)
#include "rm_device.hpp"

namespace rm {

double host_det_pow10(double y) { return det_pow10(y); }
double host_det_math(int fn, double x)
{
    switch (fn) {
    case 0: return det_log2(x);
    case 1: return det_exp2(x);
    case 2: return det_log10(x);
    case 3: return det_pow10(x);
    case 4: return det_normal(x);
    case 5: return q80_to_double(q80_from_double(x));
    default: return 0.0;
    }
}
uint64_t host_link_hash(uint64_t seed, uint32_t a, uint32_t b, double *u)
{
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    const uint64_t h = mix64(mix64(seed + 0x9E3779B97F4A7C15ull) ^ ((uint64_t(lo) << 32) | uint64_t(hi)));
    if (u) *u = (double(h >> 12) + 0.5) * 0x1.0p-52;
    return h;
}
uint64_t host_mix64(uint64_t z) { return mix64(z); }
void host_lcg_jump_map(uint64_t steps, uint64_t *A, uint64_t *C) { lcg_jump_map(steps, *A, *C); }

// Several independent ticks per launch (rm_batch_*): blockIdx.z selects the tick.  The ticks'
// descriptors sit in device memory (read with scalar loads at a uniform address); k_store_ticks
// writes them there from its kernel arguments, kStoreTicks at a time (4 KB of arguments).
constexpr int kStoreTicks = 5;
struct TickGroup {
    TickDev t[kStoreTicks];
};
static_assert(sizeof(TickGroup) + 32 <= 4096, "kernel arguments are limited to 4 KB");

__global__ void __launch_bounds__(64) k_store_ticks(const TickGroup g, TickDev *dst, int n)
{
    // one descriptor per workgroup, copied as 32-bit words
    static_assert(sizeof(TickDev) % 4 == 0, "");
    if (int(blockIdx.x) >= n) return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&g.t[blockIdx.x]);
    uint32_t *out = reinterpret_cast<uint32_t *>(dst + blockIdx.x);
    for (int i = threadIdx.x; i < int(sizeof(TickDev) / 4); i += blockDim.x) out[i] = src[i];
}

// larger batches: the descriptors are read from pinned, host-mapped memory by the device itself (no copy
// engine between two kernels of the stream)
__global__ void __launch_bounds__(64) k_fetch_ticks(const TickDev *__restrict__ host_mapped, TickDev *dst, int n)
{
    if (int(blockIdx.x) >= n) return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(host_mapped + blockIdx.x);
    uint32_t *out = reinterpret_cast<uint32_t *>(dst + blockIdx.x);
    for (int i = threadIdx.x; i < int(sizeof(TickDev) / 4); i += blockDim.x) out[i] = src[i];
}

hipError_t launch_fetch_ticks(hipStream_t s, const TickDev *host_mapped, int n, TickDev *dev_ticks)
{
    RM_KLAUNCH(k_fetch_ticks, dim3(n), dim3(64), 0, s, host_mapped, dev_ticks, n);
    return hipGetLastError();
}

// ============================================================================ per-packet call
// rm_transmit (one RadioMedium.transmit): the packet's record travels in the kernel arguments, and
// its heard links come back through ONE block of host-mapped memory (header + up to kTransmitMax
// links), so that the call is a handful of launches and one stream synchronisation -- no staging
// copies in either direction.

__global__ void __launch_bounds__(64) k_store_record(rm_tx_record r, rm_tx_record *dst)
{
    if (threadIdx.x == 0) *dst = r;
}

__global__ void __launch_bounds__(256) k_pack_result(TickDev t, TransmitResult *out)
{
    const uint32_t total = t.out_count[2];
    const uint32_t n = min(min(t.out_count[0], total), uint32_t(kTransmitMax));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out->stored = n;
        out->dropped = t.out_count[1];
        out->total = total;
        out->interference = t.pkt_interference[0];
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        out->dst[i] = t.out_dst[i];
        out->verdict[i] = t.out_verdict[i];
        out->rssi[i] = t.out_rssi[i];
        out->sinr[i] = t.out_sinr ? t.out_sinr[i] : 0.0;
    }
}

// The tick's records, offsets and per-packet flags written straight into host-mapped memory; the
// workgroup that finishes last publishes the sequence number the host waits for.
__global__ void __launch_bounds__(256)
k_pack_tick(TickDev t, int n_new, int have_offsets, HostView v, uint32_t *done_counter, uint32_t seq)
{
    const uint32_t total = t.out_count[2];
    const uint32_t n = min(min(t.out_count[0], total), v.links);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    for (uint32_t i = tid; i < n; i += step) {
        v.dst[i] = t.out_dst[i];
        v.verdict[i] = t.out_verdict[i];
        if (v.rssi) v.rssi[i] = t.out_rssi[i];     // (nullptr: the packet's transmit power, once per packet below)
        if (t.out_sinr) v.sinr[i] = t.out_sinr[i]; // no SINR extension: no sinr column crosses the link (8 of 25 bytes)
    }
    const uint32_t np = min(uint32_t(max(n_new, 0)), v.packets);
    for (uint32_t i = tid; i < np; i += step) {
        v.pkt_interference[i] = t.pkt_interference[i];
        if (!v.rssi) v.pkt_rssi[i] = t.tx[t.first_new + int(i)].txpower;
    }
    for (uint32_t i = tid; i <= np; i += step) v.pkt_offset[i] = have_offsets ? t.slot_off[t.shift + i] : 0u;
    __shared__ uint32_t s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains its stores, ...
    __syncthreads();                                  // ... the workgroup meets, one lane releases them to the host
    if (threadIdx.x == 0) {
        __threadfence_system();
        s_last = (atomicAdd(done_counter, 1u) == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        *done_counter = 0u;
        v.hdr->stored = n;
        v.hdr->dropped = t.out_count[1];
        v.hdr->total = total;
        v.hdr->span_flag = t.out_count[4];
        v.hdr->n_packets = uint32_t(max(n_new, 0));
        __threadfence_system();
        __hip_atomic_store(&v.hdr->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The same for the slots of a batch: blockIdx.y = slot; the slots' records are packed back to back
// (a slot's first record = the sum of the records of the slots before it, recomputed by every
// workgroup from the slots' counters), the packet arrays at the bases the host computed.
__global__ void __launch_bounds__(256)
k_pack_batch(const PackSlot *__restrict__ slots, int n_slots, HostView v, BatchCounts *counts, uint32_t *done_counter, uint32_t seq)
{
    __shared__ uint32_t s_part[4];
    __shared__ uint32_t s_last;
    const int b = blockIdx.y;
    uint32_t before = 0;
    for (int a = threadIdx.x; a < b; a += blockDim.x) {
        const uint32_t *oc = slots[a].t.out_count;
        before += min(oc[0], oc[2]);
    }
    for (int d = 32; d >= 1; d >>= 1) before += uint32_t(__shfl_xor(int(before), d));
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = before;
    __syncthreads();
    const uint32_t link_base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const PackSlot &ps = slots[b];
    const TickDev &t = ps.t;
    const uint32_t total = t.out_count[2];
    const uint32_t stored = min(t.out_count[0], total);
    const uint32_t n = (link_base + stored <= v.links) ? stored : 0u; // no room: the host grows the block and packs again
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    for (uint32_t i = tid; i < n; i += step) {
        const uint32_t o = link_base + i;
        v.dst[o] = t.out_dst[i];
        v.verdict[o] = t.out_verdict[i];
        if (v.rssi) v.rssi[o] = t.out_rssi[i];
        if (t.out_sinr) v.sinr[o] = t.out_sinr[i];
    }
    const uint32_t np = uint32_t(max(ps.n_new, 0));
    for (uint32_t i = tid; i < np; i += step) {
        v.pkt_interference[ps.pkt_base + i] = t.pkt_interference[i];
        if (!v.rssi) v.pkt_rssi[ps.pkt_base + i] = t.tx[t.first_new + int(i)].txpower;
    }
    for (uint32_t i = tid; i <= np; i += step) v.pkt_offset[ps.pkt_base + uint32_t(b) + i] = ps.have_offsets ? t.slot_off[t.shift + i] : 0u;
    if (tid == 0) {
        BatchCounts c{};
        c.stored = stored;
        c.dropped = t.out_count[1];
        c.total = total;
        c.span_flag = t.out_count[4];
        c.link_base = link_base;
        counts[b] = c;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains its stores, ...
    __syncthreads();                                  // ... the workgroup meets, one lane releases them to the host
    if (threadIdx.x == 0) {
        __threadfence_system();
        s_last = (atomicAdd(done_counter, 1u) == gridDim.x * gridDim.y - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        *done_counter = 0u;
        __threadfence_system();
        __hip_atomic_store(&v.hdr->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One packet in ONE launch of ONE workgroup (geometric media on a sorted table): the frame is tested
// against the boxes of 1024 receivers (one per thread), the group boxes of the near ones, then the
// receivers of the near groups (one wave per group); the few hits are evaluated exactly on the
// spot, ranked by node index in LDS, one lane walks java.util.Random over the ordered links, and
// the answer goes straight to the host-mapped block.  No cross-workgroup hand-off, no second
// launch: the per-packet latency is one kernel of a few dependent round trips.  Anything that
// does not fit the LDS lists (unbounded range, > kTransmitMax links) raises `fallback` and touches
// nothing else -- the host then takes the general path.
constexpr int kOneL1 = 1024, kOneL2 = 1024;
constexpr uint32_t kOneFallback = 0xFFFFFFFFu;

template <int MODEL>
__global__ void __launch_bounds__(1024) k_transmit_one(const NodesDev nd, const ModelDev m, const rm_tx_record tx,
                                                      uint64_t *rng_state, TransmitResult *out, uint32_t seq)
{
    __shared__ int s_l1[kOneL1], s_l2[kOneL2];
    __shared__ int s_orig[kTransmitMax], s_order[kTransmitMax];
    __shared__ double s_rssi[kTransmitMax], s_prob[kTransmitMax];
    __shared__ uint8_t s_verdict[kTransmitMax];
    __shared__ uint32_t s_n1, s_n2, s_n, s_over, s_interf;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    if (tid == 0) s_n1 = s_n2 = s_n = s_over = s_interf = 0u;
    float4 f;
    double thr64;
    tx_prefilter(m, tx, f, thr64);
    __syncthreads();
    const int n_groups = (nd.n_rx + kGroup - 1) / kGroup;
    const int n_boxes = (n_groups + 15) / 16;
    if (!(f.w < __builtin_inff())) { // no geometric bound for this frame
        if (tid == 0) s_over = 1u;
    } else if (f.w >= 0.f) {
        // level 1: boxes of 16 groups
        for (int b = tid; b < n_boxes; b += blockDim.x) {
            const float4 q = nd.wg_box_xy[b];
            const float2 qz = nd.wg_box_z[b];
            const float dx = fmaxf(fmaxf(q.x - f.x, f.x - q.z), 0.f);
            const float dy = fmaxf(fmaxf(q.y - f.y, f.y - q.w), 0.f);
            const float dz = fmaxf(fmaxf(qz.x - f.z, f.z - qz.y), 0.f);
            if (dist2_f32(dx, dy, dz) <= f.w) {
                const uint32_t k = atomicAdd(&s_n1, 1u);
                if (k < uint32_t(kOneL1)) s_l1[k] = b; else s_over = 1u;
            }
        }
    }
    __syncthreads();
    if (!s_over) {
        // level 2: the group boxes of the near ones
        const int n1 = int(s_n1);
        for (int i = tid; i < n1 * 16; i += blockDim.x) {
            const int g = s_l1[i >> 4] * 16 + (i & 15);
            if (g >= n_groups) continue;
            const float4 q = nd.bbox_xy[g];
            const float2 qz = nd.bbox_z[g];
            const float dx = fmaxf(fmaxf(q.x - f.x, f.x - q.z), 0.f);
            const float dy = fmaxf(fmaxf(q.y - f.y, f.y - q.w), 0.f);
            const float dz = fmaxf(fmaxf(qz.x - f.z, f.z - qz.y), 0.f);
            if (dist2_f32(dx, dy, dz) <= f.w) {
                const uint32_t k = atomicAdd(&s_n2, 1u);
                if (k < uint32_t(kOneL2)) s_l2[k] = g; else s_over = 1u;
            }
        }
    }
    __syncthreads();
    if (!s_over) {
        // level 3: one wave per near group; hits are evaluated with the reference's arithmetic at once
        const int n2 = uniform_i(int(s_n2));
        for (int gi = wave; gi < n2; gi += int(blockDim.x >> 6)) {
            const int j = s_l2[gi] * kGroup + lane;
            bool heard = false;
            int orig = 0;
            double rssi = 0.0, prob = 1.0;
            if (j < nd.n_rx) {
                const float4 v = nd.rxf[j];
                const float s2 = dist2_f32(v.x - f.x, v.y - f.y, v.z - f.z);
                if (s2 <= f.w && __float_as_int(v.w) == tx.channel) {
                    const RxRecord rx_ = nd.rec[j];
                    const LinkEval ev = eval_link<MODEL, false>(m, nd, tx, rx_, true);
                    if (ev.wanted) {
                        heard = true;
                        orig = rx_.orig;
                        if (MODEL == RM_MODEL_LOGDIST) {
                            rssi = ev.aux;
                            prob = rx_.rxprob;
                        } else {
                            rssi = tx.txpower;
                            prob = (MODEL == RM_MODEL_UDGM) ? ev.aux : 1.0;
                        }
                    }
                }
            }
            const uint64_t hm = ballot64(heard);
            if (hm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_n, uint32_t(__popcll(hm)));
                base = uniform_u(base);
                if (heard) {
                    const uint32_t k = base + lane_prefix(hm);
                    if (k < uint32_t(kTransmitMax)) {
                        s_orig[k] = orig;
                        s_rssi[k] = rssi;
                        s_prob[k] = prob;
                    } else {
                        s_over = 1u;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (s_over) {
        if (tid == 0) {
            out->stored = 0;
            out->dropped = 0;
            out->total = kOneFallback;
            out->interference = 0;
            __threadfence_system();
            __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // node order: rank by node index (UDGMRadioMedium.java:99 visits the node array in order)
    const int n = int(s_n);
    for (int i = tid; i < n; i += blockDim.x) {
        const int mine = s_orig[i];
        int rank = 0;
        for (int k2 = 0; k2 < n; ++k2) rank += (s_orig[k2] < mine) ? 1 : 0;
        s_order[rank] = i;
    }
    __syncthreads();
    if (tid == 0) {
        // the packet's draws, in the reference's order: Tx success first (UDGMRadioMedium.java:85-92),
        // then every heard receiver whose probability is below 1 (:106), none once the Tx failed
        uint64_t s = *rng_state & kLcgMask;
        constexpr bool kDraws = (MODEL == RM_MODEL_UDGM || MODEL == RM_MODEL_N2N || MODEL == RM_MODEL_LOGDIST);
        bool interference = false;
        if (kDraws) {
            const double txs = tx_success(m, tx);
            if (txs <= 0.0) interference = true;
            else if (txs < 1.0 && lcg_next_double(s) > txs) interference = true;
        }
        for (int r = 0; r < n; ++r) {
            const int i = s_order[r];
            uint8_t v = RM_DELIVERED;
            if (interference) {
                v = RM_INTERFERED;
            } else if (kDraws && s_prob[i] < 1.0) {
                v = (lcg_next_double(s) > s_prob[i]) ? RM_INTERFERED : RM_DELIVERED;
            }
            s_verdict[i] = v;
        }
        *rng_state = s;
        s_interf = interference ? 1u : 0u;
        out->stored = uint32_t(n);
        out->dropped = 0;
        out->total = uint32_t(n);
        out->interference = interference ? 1u : 0u;
    }
    __syncthreads();
    for (int r = tid; r < n; r += blockDim.x) {
        const int i = s_order[r];
        out->dst[r] = s_orig[i];
        out->verdict[r] = s_verdict[i];
        out->rssi[r] = s_rssi[i];
        out->sinr[r] = 0.0;
    }
    // the host polls `seq` instead of waiting for the stream: every store above is made visible
    // to the system first, then one lane publishes the sequence number
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ============================================================================ launchers

hipError_t launch_store_record(hipStream_t s, const rm_tx_record &r, rm_tx_record *dst)
{
    RM_KLAUNCH(k_store_record, dim3(1), dim3(64), 0, s, r, dst);
    return hipGetLastError();
}

hipError_t launch_transmit_one(hipStream_t s, const NodesDev &nd, const ModelDev &m, const rm_tx_record &tx,
                               uint64_t *rng_state, TransmitResult *host_mapped, uint32_t seq)
{
    const dim3 grid(1), block(1024);
    switch (m.kind) {
    case RM_MODEL_UDGM: RM_KLAUNCH(k_transmit_one<RM_MODEL_UDGM>, grid, block, 0, s, nd, m, tx, rng_state, host_mapped, seq); break;
    case RM_MODEL_UDGM_CONST:
        RM_KLAUNCH(k_transmit_one<RM_MODEL_UDGM_CONST>, grid, block, 0, s, nd, m, tx, rng_state, host_mapped, seq);
        break;
    case RM_MODEL_LOGDIST: RM_KLAUNCH(k_transmit_one<RM_MODEL_LOGDIST>, grid, block, 0, s, nd, m, tx, rng_state, host_mapped, seq); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_pack_tick(hipStream_t s, const TickDev &t, int n_new, int have_offsets, const HostView &v, uint32_t *done_counter,
                            uint32_t seq)
{
    // enough workgroups to keep the PCIe writes streaming, few enough for a short tail
    RM_KLAUNCH(k_pack_tick, dim3(64), dim3(256), 0, s, t, n_new, have_offsets, v, done_counter, seq);
    return hipGetLastError();
}

hipError_t launch_pack_batch(hipStream_t s, const PackSlot *dev_slots, int n_slots, const HostView &v, BatchCounts *host_counts,
                             uint32_t *done_counter, uint32_t seq)
{
    if (n_slots < 1 || n_slots > kMaxBatch) return hipErrorInvalidValue;
    RM_KLAUNCH(k_pack_batch, dim3(16, n_slots), dim3(256), 0, s, dev_slots, n_slots, v, host_counts, done_counter, seq);
    return hipGetLastError();
}

hipError_t launch_pack_result(hipStream_t s, const TickDev &t, TransmitResult *host_mapped)
{
    RM_KLAUNCH(k_pack_result, dim3(4), dim3(256), 0, s, t, host_mapped);
    return hipGetLastError();
}

bool batch_eligible(const TickDev &t, const LaunchCfg &cfg, const ModelDev &m)
{
    // the SINR extension is batched as well (self-contained ticks only: rm_api.cpp checks that) -- but not together
    // with java.util.Random draws: the batched SINR kernels carry no pending verdicts, such ticks take one
    // launch sequence each
    const bool sinr = m.kind == RM_MODEL_LOGDIST && (m.flags & RM_LD_SINR);
    return cfg.sorted && cfg.bbox && !cfg.f64_filter && !t.use_matrix && t.n_cnt <= kFusedScanMax &&
           t.filter_mode == kFilterWg && t.n_active > t.first_new && t.n_rx > 0 && !(sinr && cfg.stochastic);
}

hipError_t launch_store_ticks(hipStream_t s, const TickDev *ticks, int n, TickDev *dev_ticks)
{
    for (int b0 = 0; b0 < n; b0 += kStoreTicks) {
        TickGroup g{};
        const int k = min(kStoreTicks, n - b0);
        for (int i = 0; i < k; ++i) g.t[i] = ticks[b0 + i];
        RM_KLAUNCH(k_store_ticks, dim3(k), dim3(64), 0, s, g, dev_ticks + b0, k);
    }
    return hipGetLastError();
}

// rm_batch_*: n independent ticks (sorted table, fp32 frame, no SINR, <= kFusedScanMax frames each) in four
// launches (+ the descriptor upload).  stage 0: k_tick_prep + k_filter_wg, 1: k_exact, 2: k_reorder,
// 3 (SINR, between 1 and 2): k_self_entries + k_sinr.
hipError_t launch_batch_stage(hipStream_t s, int stage, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n,
                              const TickDev *b, const LaunchCfg &cfg)
{
    if (n < 1 || n > kMaxBatch) return hipErrorInvalidValue;
    if (stage == 0) return launch_filter_batch(s, nd, m, ticks, n, b, cfg);
    if (stage == 1) return launch_exact_batch(s, nd, m, ticks, n, b, cfg);
    if (stage == 3) return launch_sinr_batch(s, nd, m, ticks, n, b);
    return launch_reorder_batch(s, nd, m, ticks, n, b, cfg);
}

} // namespace rm
