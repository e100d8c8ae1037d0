// rm_filter.hip -- receiver pre-filter records, Tx packing, the all-pairs filter (grid and two-level variants)
// (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math; overview at the top of rm_engine.h)
#include "rm_device.hpp"

#include <stdlib.h>
#include <string.h>

namespace rm {

// Pre-filter record per receiver: (fx, fy, fz, channel bits) in the fp32 frame; a disabled radio
// gets a NaN position so that the geometric test can never pass (Transciever.isEnabled(),
// UDGMRadioMedium.java:102).  One wave per group of 64 receivers; the group's bounding box is the
// min/max of exactly these fp32 coordinates, so the box test is conservative w.r.t. the
// per-receiver test by monotonicity of fp32 rounding.
__global__ void __launch_bounds__(64) k_prep_rx(NodesDev nd, ModelDev m)
{
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const int i = g * kGroup + lane;
    const bool geometric = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_UDGM_CONST || m.kind == RM_MODEL_LOGDIST);
    const float nanf_ = __builtin_nanf("");
    const float inf_ = __builtin_inff();
    float4 r;
    r.x = r.y = r.z = nanf_;
    r.w = 0.f;
    if (i < nd.n_rx) {
        if (nd.enabled[i]) {
            if (geometric) {
                r.x = float(nd.x[i] - m.org_x);
                r.y = float(nd.y[i] - m.org_y);
                r.z = float(nd.z[i] - m.org_z);
            } else {
                r.x = r.y = r.z = 0.f;
            }
        }
        r.w = __int_as_float(nd.channel[i]);
        nd.rxf[i] = r;
    }
    const bool ok = (r.x == r.x);
    const float lox = wave_min(ok ? r.x : inf_), hix = wave_max(ok ? r.x : -inf_);
    const float loy = wave_min(ok ? r.y : inf_), hiy = wave_max(ok ? r.y : -inf_);
    const float loz = wave_min(ok ? r.z : inf_), hiz = wave_max(ok ? r.z : -inf_);
    // the channels heard in the group: bit (channel & 31) of every receiver that can be a candidate at all
    uint32_t chm = (ok && i < nd.n_rx) ? (1u << (uint32_t(nd.channel[min(i, nd.n_rx - 1)]) & 31u)) : 0u;
    for (int d = 32; d >= 1; d >>= 1) chm |= uint32_t(__shfl_xor(int(chm), d));
    if (lane == 0) {
        nd.bbox_xy[g] = make_float4(lox, loy, hix, hiy);
        nd.bbox_z[g] = make_float2(loz, hiz);
        nd.grp_chmask[g] = chm;
    }
}

// union of the 16 group boxes of one filter workgroup (4 waves x 4 groups = 1024 receivers)
__global__ void __launch_bounds__(256) k_wg_boxes(NodesDev nd, int n_wg)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_wg) return;
    const int n_groups = (nd.n_rx + kGroup - 1) / kGroup;
    const float inf_ = __builtin_inff();
    float4 xy = make_float4(inf_, inf_, -inf_, -inf_);
    float2 z = make_float2(inf_, -inf_);
    uint32_t chm = 0u;
    for (int g = b * 16; g < min(n_groups, b * 16 + 16); ++g) {
        const float4 q = nd.bbox_xy[g];
        const float2 qz = nd.bbox_z[g];
        chm |= nd.grp_chmask[g];
        xy.x = fminf(xy.x, q.x);
        xy.y = fminf(xy.y, q.y);
        xy.z = fmaxf(xy.z, q.z);
        xy.w = fmaxf(xy.w, q.w);
        z.x = fminf(z.x, qz.x);
        z.y = fmaxf(z.y, qz.y);
    }
    nd.wg_box_xy[b] = xy;
    nd.wg_box_z[b] = z;
    nd.wg_chmask[b] = chm;
}

// Changed nodes written in place: the source table by node index, the receiver table (SoA arrays
// and the exact-path record) at the node's engine position.  The engine order stays as it is --
// it only has to be a permutation; k_prep_rx recomputes the pre-filter records and boxes afterwards.
__global__ void __launch_bounds__(256) k_patch_nodes(NodesDev nd, const NodePatch *list, int n, NodePatch one)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const NodePatch p = list ? list[i] : one;
    const_cast<double *>(nd.sx)[p.node] = p.x;
    const_cast<double *>(nd.sy)[p.node] = p.y;
    const_cast<double *>(nd.sz)[p.node] = p.z;
    const_cast<double *>(nd.stxpower)[p.node] = p.txpower;
    const_cast<double *>(nd.stxprob)[p.node] = p.txprob;
    const_cast<double *>(nd.srxprob)[p.node] = p.rxprob;
    const_cast<int32_t *>(nd.schannel)[p.node] = p.channel;
    const_cast<uint8_t *>(nd.senabled)[p.node] = uint8_t(p.enabled);
    SrcRecord *sr = const_cast<SrcRecord *>(nd.srec) + p.node;
    sr->x = p.x;
    sr->y = p.y;
    sr->z = p.z;
    sr->txpower = p.txpower;
    sr->txprob = p.txprob;
    sr->channel = p.channel;
    if (p.pos < 0) return;
    const_cast<double *>(nd.x)[p.pos] = p.x;
    const_cast<double *>(nd.y)[p.pos] = p.y;
    const_cast<double *>(nd.z)[p.pos] = p.z;
    const_cast<double *>(nd.rxprob)[p.pos] = p.rxprob;
    const_cast<int32_t *>(nd.channel)[p.pos] = p.channel;
    const_cast<uint8_t *>(nd.enabled)[p.pos] = uint8_t(p.enabled);
    RxRecord *r = const_cast<RxRecord *>(nd.rec) + p.pos;
    r->x = p.x;
    r->y = p.y;
    r->z = p.z;
    r->rxprob = p.rxprob;
    r->channel = p.channel;
    r->enabled = p.enabled;
    if (!nd.rec32) return;
    RxCompact *c = const_cast<RxCompact *>(nd.rec32) + p.pos;
    c->x = p.x;
    c->y = p.y;
    c->z = p.z;
    c->flags = (p.rxprob != 1.0) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
k_pack_tx(NodesDev nd, const int32_t *src, int n, int64_t start_us, int64_t air_us, rm_tx_record *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rm_tx_record r = make_tx_record(nd, src[i], start_us, air_us);
    out[i] = r;
}

constexpr int kPackChunk = 128; // ticks per k_pack_tx_batch launch: their start times travel in the kernel arguments (1 KB)
struct PackStarts {
    int64_t start_us[kPackChunk];
};

// (blockIdx.z: the rank whose [n_ticks][n] block of source indices this is -- one block when a rank packs its own
// transmitters, `world` of them when the INDICES were all-gathered and every rank builds all records itself: 4 bytes per
// frame over the links between the GPUs instead of 64)
__global__ void __launch_bounds__(256)
k_pack_tx_batch(NodesDev nd, const int32_t *src, int n, PackStarts st, int64_t air_us, rm_tx_record *out, int tick0, int n_ticks)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t o = (size_t(blockIdx.z) * n_ticks + size_t(tick0 + blockIdx.y)) * n + i;
    const rm_tx_record r = make_tx_record(nd, src[o], st.start_us[blockIdx.y], air_us);
    out[o] = r;
}

template <int RPT, bool F64, bool BBOX, bool SHADOW>
__global__ void __launch_bounds__(kBlock, F64 ? 2 : 6) k_filter(const NodesDev nd, const ModelDev m, const TickDev t)
{
    __shared__ float4 s_txf[kTxChunk];
    __shared__ int s_ch[kTxChunk];
    __shared__ double s_td[F64 ? kTxChunk * 4 : 1];
    __shared__ uint64_t s_mask[kWavesPerBlock][kTxChunk][RPT]; // candidate ballots of the near frames
    __shared__ uint32_t s_tbl[SHADOW ? kShadowBins : 1];
    __shared__ float s_inv[SHADOW ? kTxChunk : 1]; // bins / thr of the frame (0: table not usable for it)
    __shared__ int s_src[SHADOW ? kTxChunk : 1];

    const int lane = threadIdx.x & 63;
    const int wave = wave_index();
    const int slab = blockIdx.x * kWavesPerBlock + wave;
    const int chunk = blockIdx.y;
    const int n_eval = t.n_active - t.first_eval;
    const int e0 = chunk * kTxChunk; // eval-relative index of the tile's first frame
    const int nt = min(kTxChunk, n_eval - e0);
    const int jbase = slab * (kGroup * RPT);

    // the next tick's counters (other parity) are zeroed here: nothing touches them during this tick
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (threadIdx.x < 8) t.next_counters[threadIdx.x] = 0u;
        t.next_shard_count[threadIdx.x * kShardStride] = 0u; // kBlock == kShards
    }

    // receivers of this lane (coalesced 16-byte loads), resident in registers for the whole tile;
    // issued before the tile is staged so that both round trips overlap
    float fx[RPT], fy[RPT], fz[RPT];
    int fch[RPT];
    int forig[RPT];
    double gx[RPT], gy[RPT], gz[RPT];
    float4 bxy[RPT];
    if (SHADOW) s_tbl[threadIdx.x] = m.shadow_tbl[threadIdx.x]; // kBlock == kShadowBins
    float2 bz[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int j = jbase + r * kGroup + lane;
        fx[r] = fy[r] = fz[r] = __builtin_nanf("");
        fch[r] = 0;
        forig[r] = 0;
        if (F64) gx[r] = gy[r] = gz[r] = u2f(0x7FF8000000000000ull);
        if (j < t.n_rx) {
            const float4 v = nd.rxf[j];
            fx[r] = v.x;
            fy[r] = v.y;
            fz[r] = v.z;
            fch[r] = __float_as_int(v.w);
            if (SHADOW) forig[r] = nd.orig[j];
            if (F64 && v.x == v.x) {
                gx[r] = nd.x[j];
                gy[r] = nd.y[j];
                gz[r] = nd.z[j];
            }
        }
        if (BBOX) {
            const int g = slab * RPT + r;
            const bool ok = g * kGroup < t.n_rx;
            bxy[r] = ok ? nd.bbox_xy[g] : make_float4(0.f, 0.f, 0.f, 0.f);
            bz[r] = ok ? nd.bbox_z[g] : make_float2(0.f, 0.f);
        }
    }

    // stage the transmitter tile in LDS: one frame per lane of wave 0, pre-filter record computed
    // on the fly from the on-air record
    if (threadIdx.x < kTxChunk) {
        float4 f = make_float4(0.f, 0.f, 0.f, -1.f);
        double thr64 = -1.0;
        int ch = 0;
        int src_id = -1;
        double px = 0, py = 0, pz = 0;
        if (int(threadIdx.x) < nt) {
            rm_tx_record tx;
            const int abs_i = t.first_eval + e0 + int(threadIdx.x);
            if (t.src_list && abs_i >= t.first_new) { // build mode: a new frame's record comes from the source table
                tx = make_tx_record(nd, t.src_list[abs_i - t.first_new], t.src_start_us, t.src_air_us);
                if (blockIdx.x == 0) t.tx_build[abs_i] = tx;
            } else { // a frame already on the air (or records given by the caller)
                tx = t.tx[abs_i];
                if (t.check_txprob && blockIdx.x == 0 && tx.src >= 0 && tx.txprob > 0.0 && tx.txprob < 1.0) t.stage_count[6] = 2u;
            }
            tx_prefilter(m, tx, f, thr64);
            ch = tx.channel;
            src_id = tx.src;
            px = tx.x;
            py = tx.y;
            pz = tx.z;
        }
        s_txf[threadIdx.x] = f;
        s_ch[threadIdx.x] = ch;
        if (SHADOW) {
            // the table is indexed by rho = s2 / thr; usable if the fp32 frame error is small against
            // the distances where it decides anything (d > 0.15 cut), else bin 0 (always pass)
            float inv = 0.f;
            if (f.w > 0.f && f.w < __builtin_inff()) {
                const double cut = sqrt(double(f.w));
                if (2.0 * m.f32_slack / (0.15 * cut) + 1e-5 <= kShadowPad) inv = float(kShadowBins) / f.w;
            }
            s_inv[threadIdx.x] = inv;
            s_src[threadIdx.x] = src_id;
        }
        if (F64) {
            s_td[threadIdx.x * 4 + 0] = px;
            s_td[threadIdx.x * 4 + 1] = py;
            s_td[threadIdx.x * 4 + 2] = pz;
            s_td[threadIdx.x * 4 + 3] = thr64;
        }
    }
    __syncthreads();
    // counters that later kernels of this tick (cursor) or the next tick's filter (candidate
    // totals, other parity) add to start at zero.  Every thread of the x = 0 workgroups takes part,
    // also the waves without a slab (a table of fewer than four slabs): they leave only afterwards.
    if (!t.use_matrix && blockIdx.x == 0) {
        for (int i = blockIdx.y * kBlock + threadIdx.x; i < t.zero_len; i += gridDim.y * kBlock) {
            t.cursor[i] = 0u;
            t.cand_tot_next[i] = 0u;
        }
    }
    if (slab >= t.n_slabs) return;
    if (t.use_matrix) {
        if (e0 >= t.cnt_base) t.cnt[(size_t((e0 - t.cnt_base) / kTxChunk) * t.n_slabs + slab) * 64 + lane] = 0u;
    }

    // which frames of the tile can reach which receiver group: one frame per lane against the
    // group's bounding box, one ballot per group
    uint64_t near[RPT];
    uint64_t todo = 0;
    if (BBOX) {
        const float4 tf = s_txf[lane];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            near[r] = 0;
            if ((slab * RPT + r) * kGroup < t.n_rx) {
                const float dx = fmaxf(fmaxf(bxy[r].x - tf.x, tf.x - bxy[r].z), 0.f);
                const float dy = fmaxf(fmaxf(bxy[r].y - tf.y, tf.y - bxy[r].w), 0.f);
                const float dz = fmaxf(fmaxf(bz[r].x - tf.z, tf.z - bz[r].y), 0.f);
                near[r] = ballot64(dist2_f32(dx, dy, dz) <= tf.w);
            }
            todo |= near[r];
        }
    } else {
        todo = (nt >= 64) ? ~0ull : ((1ull << nt) - 1ull);
#pragma unroll
        for (int r = 0; r < RPT; ++r) near[r] = todo;
    }

    // pass 1: per near frame, the candidate ballots of the RPT groups; lane ti keeps frame ti's count
    uint32_t my_total = 0;
    uint64_t walk = todo;
    while (walk) {
        const int ti = __ffsll((long long)walk) - 1; // wave-uniform
        walk &= walk - 1;
        const float4 tf = s_txf[ti];
        const int tch = s_ch[ti];
        uint64_t mask[RPT];
        uint32_t total = 0;
        if (F64) {
            const double px = s_td[ti * 4 + 0], py = s_td[ti * 4 + 1], pz = s_td[ti * 4 + 2], thr = s_td[ti * 4 + 3];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const double dx = px - gx[r], dy = py - gy[r], dz = pz - gz[r];
                const double s2 = dx * dx + dy * dy + dz * dz;
                mask[r] = ballot64((s2 <= thr) && (fch[r] == tch));
                total += uint32_t(__popcll(mask[r]));
            }
        } else {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                mask[r] = 0;
                if ((near[r] >> ti) & 1ull) { // wave-uniform
                    const float s2 = dist2_f32(fx[r] - tf.x, fy[r] - tf.y, fz[r] - tf.z);
                    bool hit = (s2 <= tf.w) && (fch[r] == tch);
                    if (SHADOW && hit) {
                        // second level: with this link's shadowing deviate, can it still reach the
                        // level?  Conservative table of the largest hash that can, per bin of d^2/cut^2.
                        const int bin = min(kShadowBins - 1, int(s2 * s_inv[ti]));
                        const uint32_t a = uint32_t(s_src[ti]), b = uint32_t(forig[r]);
                        const uint64_t key = (uint64_t(a < b ? a : b) << 32) | uint64_t(a < b ? b : a);
                        hit = uint32_t(mix64(m.ld_seed_mixed ^ key) >> 32) <= s_tbl[bin];
                    }
                    mask[r] = ballot64(hit);
                    total += uint32_t(__popcll(mask[r]));
                }
            }
        }
        if (total) {
            if (lane == ti) my_total = total;
            if (lane < RPT) {
                uint64_t v = mask[0];
#pragma unroll
                for (int r = 1; r < RPT; ++r) v = (lane == r) ? mask[r] : v;
                s_mask[wave][ti][lane] = v;
            }
        }
    }
    const uint64_t have = ballot64(my_total != 0u);
    if (have == 0) return; // the common case: far from every transmitter of the tile

    // candidate links per frame (frames that get verdicts only): sizes the frame's segment
    if (!t.use_matrix && my_total != 0u && t.first_eval + e0 + lane >= t.first_new)
        atomicAdd(&t.cand_tot[e0 + lane - t.cnt_base], my_total);

    // one atomic reserves the contiguous run of candidate entries of this (tile, slab); the frames'
    // blocks follow each other inside it in frame order
    uint32_t inc = my_total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    const uint32_t wave_total = __shfl(inc, 63);
    const uint32_t shard = (blockIdx.x + blockIdx.y * gridDim.x) & t.shard_mask;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&t.shard_count[shard * kShardStride], wave_total);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base + wave_total > t.seg_cap) { // the shard is full: drop the run, flag the tick
        if (lane == 0) t.stage_count[1] = 1u;
        return;
    }
    const uint32_t my_base = shard * t.seg_cap + base + inc - my_total;

    // pass 2: fill the blocks in receiver order
    walk = have;
    while (walk) {
        const int ti = __ffsll((long long)walk) - 1;
        walk &= walk - 1;
        const uint32_t fbase = __shfl(my_base, ti);
        uint32_t pre = 0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const uint64_t mk = s_mask[wave][ti][r];
            if (mk == 0) continue;
            if ((mk >> lane) & 1ull) {
                const uint32_t idx = fbase + pre + lane_prefix(mk);
                t.st_pkt[idx] = e0 + ti;
                t.st_dst[idx] = jbase + r * kGroup + lane;
                if (t.use_matrix) t.st_blk[idx] = fbase; // the run's base: only the ordered scatter of unsorted tables ranks inside it
            }
            pre += uint32_t(__popcll(mk));
        }
    }
}

// ============================================================================ two-level filter
// k_tick_prep: one thread per swept frame -- builds the frame's on-air record (build mode) and its
// pre-filter record once per tick, and zeroes the counters later kernels add to.
// k_filter_wg: one workgroup per 4*RPT receiver groups.  Phase A: every thread tests frames
// against the union box of the workgroup's receivers and the near ones are compacted into LDS
// (a few dozen of a thousand at the bench densities).  Phase B: each wave runs the two-pass
// filter of k_filter over chunks of 64 near frames for its own RPT groups.

constexpr int kNearLds = 512; // near-frame records held in LDS between two phase-B rounds

constexpr int kRfThreads = 256;
RM_D float prefilter_inv(const ModelDev &m, const float4 &f) // (bins / thr of the frame for the shadowed medium's table, 0: not usable)
{
    float inv = 0.f;
    if (m.shadow_tbl && f.w > 0.f && f.w < __builtin_inff()) {
        const double cut = sqrt(double(f.w));
        if (2.0 * m.f32_slack / (0.15 * cut) + 1e-5 <= kShadowPad) inv = float(kShadowBins) / f.w;
    }
    return inv;
}

RM_D void tick_prep_body(const NodesDev &nd, const ModelDev &m, const TickDev &t)
{
    const int n_eval = t.n_active - t.first_eval;
    if (blockIdx.x == 0) {
        if (threadIdx.x < 8) t.next_counters[threadIdx.x] = 0u;
        t.next_shard_count[threadIdx.x * kShardStride] = 0u; // kBlock == kShards
    }
    if (!t.use_matrix)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < t.zero_len; i += gridDim.x * blockDim.x) {
            t.cursor[i] = 0u;
            t.cand_tot_next[i] = 0u;
        }
    if (t.reset_heads) // SINR tick of a batch: every receiver's link list starts empty
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < t.n_rx; i += gridDim.x * blockDim.x) t.head[i] = -1;
    if (t.acc_lo) // ... or, summed per receiver (TickDev::acc_lo): every receiver's sum starts at zero, nobody is on the air yet
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < t.n_rx; i += gridDim.x * blockDim.x) {
            t.acc_lo[i] = 0ull;
            t.acc_hi[i] = 0ull;
        }
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (t.air.pool != nullptr) { // block-uniform.  SINR with the lists that live across ticks:
        if (blockIdx.x == 0 && threadIdx.x == 0 && t.air.bad[0]) t.stage_count[1] = 1u; // the lists are broken until the host rebuilds them
        // half duplex: every swept frame leaves a SELF entry in its source's list (whole waves allocate together)
        bool want = false;
        int pos = 0;
        rm_tx_record txs{};
        if (e < n_eval) {
            const int ai = t.first_eval + e;
            txs = (t.src_list && ai >= t.first_new) ? make_tx_record(nd, t.src_list[ai - t.first_new], t.src_start_us, t.src_air_us) : t.tx[ai];
            pos = engine_pos(nd, txs.src);
            want = pos >= 0;
        }
        const int aidx = air_alloc(t, want, air_sub(t));
        if (want) air_link(t, aidx, pos, txs.start_us, txs.air_us, 0.0, kAirSelf);
    }
    if (e >= n_eval) return;
    const int abs_i = t.first_eval + e;
    rm_tx_record tx;
    if (t.src_list && abs_i >= t.first_new) {
        tx = make_tx_record(nd, t.src_list[abs_i - t.first_new], t.src_start_us, t.src_air_us);
        t.tx_build[abs_i] = tx;
    } else if (t.gather_idx) { // (batches only: first_eval == first_new == 0) the gathered source indices: the record is built here
        tx = make_tx_record(nd, t.gather_idx[size_t(abs_i / t.gather_slots) * size_t(t.gather_stride) + size_t(abs_i % t.gather_slots)],
                            t.src_start_us, t.src_air_us);
        t.tx_build[abs_i] = tx;
    } else if (t.gather_src) { // (batches only) gathered records
        tx = t.gather_src[size_t(abs_i / t.gather_slots) * size_t(t.gather_stride) + size_t(abs_i % t.gather_slots)];
        t.tx_build[abs_i] = tx;
    } else {
        tx = t.tx[abs_i];
    }
    if (t.check_span && tx.src >= 0 && (tx.start_us < t.span_begin || tx.start_us + tx.air_us > t.span_end))
        t.stage_count[6] = 1u; // reported as RM_ERR_STATE when the tick's result is read
    if (t.check_txprob && tx.src >= 0 && tx.txprob > 0.0 && tx.txprob < 1.0) t.stage_count[6] = 2u; // (as well)
    float4 f;
    double thr64;
    tx_prefilter(m, tx, f, thr64);
    const float inv = prefilter_inv(m, f);
    t.p_txf[e] = f;
    t.p_ch[e] = tx.channel;
    t.p_src[e] = tx.src;
    t.p_inv[e] = inv;
}

__global__ void __launch_bounds__(256) k_tick_prep(const NodesDev nd, const ModelDev m, const TickDev t)
{
    tick_prep_body(nd, m, t);
}

// A filter workgroup's receivers: one per lane in each of the wave's RPT groups, with the groups' boxes and the union box
// of the workgroup's 4 * RPT groups.  They stay in registers for all the ticks the workgroup sweeps (filter_wg_body).
template <int RPT> struct WgRx {
    float fx[RPT], fy[RPT], fz[RPT];
    int fch[RPT], forig[RPT];
    float4 bxy[RPT];
    float2 bz[RPT];
    uint32_t bmask[RPT]; // the groups' channel masks
    float4 wxy;
    float2 wz;
    uint32_t wmask;      // ... and the workgroup's
};

template <int RPT, bool SHADOW>
RM_D void wg_rx_load(const NodesDev &nd, const TickDev &t, WgRx<RPT> &rx)
{
    const int lane = threadIdx.x & 63;
    const int wave = wave_index();
    const int wg = blockIdx.x;
    const int slab = wg * kWavesPerBlock + wave;
    const int jbase = slab * (kGroup * RPT);
    const bool live = slab < t.n_slabs;
    const int n_groups = (t.n_rx + kGroup - 1) / kGroup;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int j = jbase + r * kGroup + lane;
        rx.fx[r] = rx.fy[r] = rx.fz[r] = __builtin_nanf("");
        rx.fch[r] = 0;
        rx.forig[r] = 0;
        if (live && j < t.n_rx) {
            const float4 v = nd.rxf[j];
            rx.fx[r] = v.x;
            rx.fy[r] = v.y;
            rx.fz[r] = v.z;
            rx.fch[r] = __float_as_int(v.w);
            if (SHADOW) rx.forig[r] = nd.orig[j];
        }
        const int g = slab * RPT + r;
        const bool ok = live && g * kGroup < t.n_rx;
        rx.bxy[r] = ok ? nd.bbox_xy[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        rx.bz[r] = ok ? nd.bbox_z[g] : make_float2(0.f, 0.f);
        rx.bmask[r] = ok ? nd.grp_chmask[g] : 0u;
    }
    // union box of the workgroup's 4*RPT groups
    if (RPT == 4) {
        rx.wxy = nd.wg_box_xy[wg];
        rx.wz = nd.wg_box_z[wg];
        rx.wmask = nd.wg_chmask[wg];
    } else {
        const float inf_ = __builtin_inff();
        rx.wxy = make_float4(inf_, inf_, -inf_, -inf_);
        rx.wz = make_float2(inf_, -inf_);
        rx.wmask = 0u;
        for (int g = wg * kWavesPerBlock * RPT; g < min(n_groups, (wg + 1) * kWavesPerBlock * RPT); ++g) { // uniform
            const float4 q = nd.bbox_xy[g];
            const float2 qz = nd.bbox_z[g];
            rx.wmask |= nd.grp_chmask[g];
            rx.wxy.x = fminf(rx.wxy.x, q.x);
            rx.wxy.y = fminf(rx.wxy.y, q.y);
            rx.wxy.z = fmaxf(rx.wxy.z, q.z);
            rx.wxy.w = fmaxf(rx.wxy.w, q.w);
            rx.wz.x = fminf(rx.wz.x, qz.x);
            rx.wz.y = fmaxf(rx.wz.y, qz.y);
        }
    }
}

// one tick of the two-level filter for the receivers in `rx`; tick_salt spreads the ticks of a launch over the shards
template <int RPT, bool SHADOW>
RM_D void filter_wg_tick(const NodesDev &nd, const ModelDev &m, const TickDev &t, const WgRx<RPT> &rx, const bool first_of_wg,
                         const uint32_t tick_salt)
{
    __shared__ float4 s_txf[kNearLds];
    __shared__ int s_ch[kNearLds];
    __shared__ int s_e[kNearLds];
    __shared__ float s_inv[SHADOW ? kNearLds : 1];
    __shared__ int s_src[SHADOW ? kNearLds : 1];
    __shared__ uint64_t s_mask[kWavesPerBlock][kTxChunk][RPT];
    __shared__ uint32_t s_tbl[SHADOW ? kShadowBins : 1];
    __shared__ uint32_t s_n;

    const int lane = threadIdx.x & 63;
    const int wave = wave_index();
    const int wg = blockIdx.x;
    const int slab = wg * kWavesPerBlock + wave;
    const int jbase = slab * (kGroup * RPT);
    const bool live = slab < t.n_slabs;
    const int n_eval = t.n_active - t.first_eval;

    if (first_of_wg && SHADOW) s_tbl[threadIdx.x] = m.shadow_tbl[threadIdx.x];
    if (threadIdx.x == 0) s_n = 0u;

    // phase A works on kUnrollA x 256 frames at a time: their records are requested together (one
    // exposed round trip per 1024 frames), the first ones before anything else
    constexpr int kUnrollA = 4;
    float4 af[kUnrollA];
    int ach[kUnrollA], asrc[kUnrollA], ae[kUnrollA];
    float ainv[kUnrollA];
    // the frames phase A looks at: all of the tick's, or (RPT == 4, large tables in batches) the list of those near this
    // workgroup's block of kNearSb workgroups
    const int32_t *const near_list = (RPT == 4 && t.near_list) ? t.near_list + size_t(wg / kNearSb) * size_t(t.near_cap) : nullptr;
    const int n_look = near_list ? uniform_i(int(min(t.near_cnt[wg / kNearSb], uint32_t(t.near_cap)))) : n_eval;
    auto request = [&](int g0) {
#pragma unroll
        for (int u = 0; u < kUnrollA; ++u) {
            const int i = g0 + u * kBlock + int(threadIdx.x);
            af[u] = make_float4(0.f, 0.f, 0.f, -1.f);
            ach[u] = 0;
            asrc[u] = -1;
            ainv[u] = 0.f;
            ae[u] = -1;
            if (i < n_look) {
                const int e = near_list ? near_list[i] : i;
                ae[u] = e;
                af[u] = t.p_txf[e];
                ach[u] = t.p_ch[e];
                if (SHADOW) {
                    ainv[u] = t.p_inv[e];
                    asrc[u] = t.p_src[e];
                }
            }
        }
    };
    request(0);
    const float (&fx)[RPT] = rx.fx, (&fy)[RPT] = rx.fy, (&fz)[RPT] = rx.fz;
    const int (&fch)[RPT] = rx.fch, (&forig)[RPT] = rx.forig;
    const float4 (&bxy)[RPT] = rx.bxy;
    const float2 (&bz)[RPT] = rx.bz;
    const float4 wxy = rx.wxy;
    const float2 wz = rx.wz;
    const uint32_t (&bmask)[RPT] = rx.bmask;
    const uint32_t wmask = rx.wmask;
    __syncthreads();

    uint32_t round = 0;
    for (int g0 = 0; g0 < n_look; g0 += kUnrollA * kBlock) { // block-uniform
    if (g0) request(g0);
#pragma unroll 1
    for (int u = 0; u < kUnrollA; ++u) { // rolled: one copy of phase B; the records are selected, not indexed
        const int f0 = g0 + u * kBlock;
        if (f0 >= n_look) break; // block-uniform
        // phase A: this thread's frame against the workgroup box
        float4 tfa = af[0];
        int cha = ach[0], srca = asrc[0], e = ae[0];
        float inva = ainv[0];
#pragma unroll
        for (int k = 1; k < kUnrollA; ++k) {
            tfa.x = (u == k) ? af[k].x : tfa.x;
            tfa.y = (u == k) ? af[k].y : tfa.y;
            tfa.z = (u == k) ? af[k].z : tfa.z;
            tfa.w = (u == k) ? af[k].w : tfa.w;
            cha = (u == k) ? ach[k] : cha;
            srca = (u == k) ? asrc[k] : srca;
            inva = (u == k) ? ainv[k] : inva;
            e = (u == k) ? ae[k] : e;
        }
        bool hit = false;
        if (e >= 0) {
            const float dx = fmaxf(fmaxf(wxy.x - tfa.x, tfa.x - wxy.z), 0.f);
            const float dy = fmaxf(fmaxf(wxy.y - tfa.y, tfa.y - wxy.w), 0.f);
            const float dz = fmaxf(fmaxf(wz.x - tfa.z, tfa.z - wz.y), 0.f);
            hit = dist2_f32(dx, dy, dz) <= tfa.w && ((wmask >> (uint32_t(cha) & 31u)) & 1u) != 0u; // (nobody here listens on its channel)
        }
        const uint64_t hm = ballot64(hit);
        if (hm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_n, uint32_t(__popcll(hm)));
            base = __builtin_amdgcn_readfirstlane(base);
            if (hit) {
                const uint32_t k = base + lane_prefix(hm);
                s_txf[k] = tfa;
                s_ch[k] = cha;
                s_e[k] = e;
                if (SHADOW) {
                    s_inv[k] = inva;
                    s_src[k] = srca;
                }
            }
        }
        __syncthreads();
        const int n_near = uniform_i(int(s_n));
        const bool last = f0 + kBlock >= n_look;
        if (!last && n_near + kBlock <= kNearLds) continue; // room for another 256 frames

        // phase B: chunks of 64 near frames, every wave for its own groups
        if (live) {
            for (int c0 = 0; c0 < n_near; c0 += kTxChunk) {
                const int nt = min(kTxChunk, n_near - c0);
                uint64_t near[RPT];
                uint64_t todo = 0;
                {
                    const float4 tf = s_txf[c0 + min(lane, nt - 1)];
                    const uint32_t tchb = uint32_t(s_ch[c0 + min(lane, nt - 1)]) & 31u;
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        near[r] = 0;
                        if ((slab * RPT + r) * kGroup < t.n_rx) {
                            const float dx = fmaxf(fmaxf(bxy[r].x - tf.x, tf.x - bxy[r].z), 0.f);
                            const float dy = fmaxf(fmaxf(bxy[r].y - tf.y, tf.y - bxy[r].w), 0.f);
                            const float dz = fmaxf(fmaxf(bz[r].x - tf.z, tf.z - bz[r].y), 0.f);
                            near[r] = ballot64(lane < nt && dist2_f32(dx, dy, dz) <= tf.w && ((bmask[r] >> tchb) & 1u) != 0u);
                        }
                        todo |= near[r];
                    }
                }
                uint32_t my_total = 0;
                uint64_t walk = todo;
                while (walk) {
                    const int ti = __ffsll((long long)walk) - 1; // wave-uniform
                    walk &= walk - 1;
                    const float4 tf = s_txf[c0 + ti];
                    const int tch = s_ch[c0 + ti];
                    uint64_t mask[RPT];
                    uint32_t total = 0;
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        mask[r] = 0;
                        if ((near[r] >> ti) & 1ull) {
                            const float s2 = dist2_f32(fx[r] - tf.x, fy[r] - tf.y, fz[r] - tf.z);
                            bool h = (s2 <= tf.w) && (fch[r] == tch);
                            if (SHADOW && h) {
                                const int bin = min(kShadowBins - 1, int(s2 * s_inv[c0 + ti]));
                                const uint32_t a = uint32_t(s_src[c0 + ti]), bb = uint32_t(forig[r]);
                                const uint64_t key = (uint64_t(a < bb ? a : bb) << 32) | uint64_t(a < bb ? bb : a);
                                h = uint32_t(mix64(m.ld_seed_mixed ^ key) >> 32) <= s_tbl[bin];
                            }
                            mask[r] = ballot64(h);
                            total += uint32_t(__popcll(mask[r]));
                        }
                    }
                    if (total) {
                        if (lane == ti) my_total = total;
                        if (lane < RPT) {
                            uint64_t v = mask[0];
#pragma unroll
                            for (int r = 1; r < RPT; ++r) v = (lane == r) ? mask[r] : v;
                            s_mask[wave][ti][lane] = v;
                        }
                    }
                }
                const uint64_t have = ballot64(my_total != 0u);
                if (have == 0) continue;
                const int my_e = s_e[c0 + min(lane, nt - 1)];
                if (!t.use_matrix && my_total != 0u && t.first_eval + my_e >= t.first_new)
                    atomicAdd(&t.cand_tot[my_e - t.cnt_base], my_total);
                uint32_t inc = my_total;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(inc, d);
                    if (lane >= d) inc += o;
                }
                const uint32_t wave_total = __shfl(inc, 63);
                const uint32_t shard = (uint32_t(slab) + (round + uint32_t(c0 >> 6)) * 37u + tick_salt * 101u) & t.shard_mask;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&t.shard_count[shard * kShardStride], wave_total);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base + wave_total > t.seg_cap) { // the shard is full: drop the run, flag the tick
                    if (lane == 0) t.stage_count[1] = 1u;
                    continue;
                }
                const uint32_t my_base = shard * t.seg_cap + base + inc - my_total;
                walk = have;
                while (walk) {
                    const int ti = __ffsll((long long)walk) - 1;
                    walk &= walk - 1;
                    const uint32_t fbase = __shfl(my_base, ti);
                    const int e_ti = s_e[c0 + ti];
                    uint32_t pre = 0;
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        const uint64_t mk = s_mask[wave][ti][r];
                        if (mk == 0) continue;
                        if ((mk >> lane) & 1ull) {
                            const uint32_t idx = fbase + pre + lane_prefix(mk);
                            t.st_pkt[idx] = e_ti;
                            t.st_dst[idx] = jbase + r * kGroup + lane;
                            if (t.use_matrix) t.st_blk[idx] = fbase; // the run's base: only the ordered scatter of unsorted tables ranks inside it
                        }
                        pre += uint32_t(__popcll(mk));
                    }
                }
            }
        }
        round += uint32_t(kNearLds / kTxChunk);
        if (!last) {
            __syncthreads(); // every wave is done with the LDS records
            if (threadIdx.x == 0) s_n = 0u;
            __syncthreads();
        }
    }
    }
    __syncthreads(); // (the next tick of this workgroup reuses the LDS lists)
}

// `ticks[first .. first + count)`: the ticks this workgroup sweeps one after the other against ITS receivers, which stay
// in registers (and their boxes) for all of them -- at a million receivers the 16 MB of pre-filter records then leave HBM
// once per `count` ticks instead of once per tick.
template <int RPT, bool SHADOW>
RM_D void filter_wg_body(const NodesDev &nd, const ModelDev &m, const TickDev *__restrict__ ticks, const int first, const int count)
{
    WgRx<RPT> rx;
    wg_rx_load<RPT, SHADOW>(nd, ticks[first], rx); // (the receiver tiling is the same for every tick of a launch)
    for (int b = 0; b < count; ++b) filter_wg_tick<RPT, SHADOW>(nd, m, ticks[first + b], rx, b == 0, uint32_t(first + b));
}

template <int RPT, bool SHADOW>
__global__ void __launch_bounds__(kBlock, RPT == 4 ? 4 : (RPT == 2 ? 5 : 6)) k_filter_wg(const NodesDev nd, const ModelDev m, const TickDev t)
{
    filter_wg_body<RPT, SHADOW>(nd, m, &t, 0, 1);
}

__global__ void __launch_bounds__(256) k_tick_prep_batch(const NodesDev nd, const ModelDev m, const TickDev *__restrict__ ticks)
{
    tick_prep_body(nd, m, ticks[blockIdx.z]);
}

// The near-frame lists of a batch over a large table (blockIdx.x = block of kNearSb filter workgroups, blockIdx.z = tick): at
// a million receivers a tick has a thousand filter workgroups, and every one of them reading and testing every frame of the
// tick was 27 MB of L2 reads per tick and a third of the filter's time; a block's box (16 k receivers) is near to a few
// dozen of a thousand frames, and its workgroups look at those.
__global__ void __launch_bounds__(256) k_near_lists(const NodesDev nd, const TickDev *__restrict__ ticks, const int n_wg)
{
    __shared__ float s_box[6];
    __shared__ uint32_t s_mask, s_n;
    const TickDev &t = ticks[blockIdx.z];
    const int sb = blockIdx.x, lane = threadIdx.x & 63;
    const int n_eval = t.n_active - t.first_eval;
    if (t.near_list == nullptr) return;
    if (threadIdx.x < 64) { // the union of the block's workgroup boxes and channel masks
        const float inf_ = __builtin_inff();
        const int w = sb * kNearSb + lane;
        const bool ok = lane < kNearSb && w < n_wg;
        const float4 q = ok ? nd.wg_box_xy[w] : make_float4(inf_, inf_, -inf_, -inf_);
        const float2 qz = ok ? nd.wg_box_z[w] : make_float2(inf_, -inf_);
        float x0 = q.x, y0 = q.y, x1 = q.z, y1 = q.w, z0 = qz.x, z1 = qz.y;
        uint32_t mk = ok ? nd.wg_chmask[w] : 0u;
        for (int d = 8; d >= 1; d >>= 1) {
            x0 = fminf(x0, __shfl_xor(x0, d));
            y0 = fminf(y0, __shfl_xor(y0, d));
            z0 = fminf(z0, __shfl_xor(z0, d));
            x1 = fmaxf(x1, __shfl_xor(x1, d));
            y1 = fmaxf(y1, __shfl_xor(y1, d));
            z1 = fmaxf(z1, __shfl_xor(z1, d));
            mk |= uint32_t(__shfl_xor(int(mk), d));
        }
        if (threadIdx.x == 0) {
            s_box[0] = x0, s_box[1] = y0, s_box[2] = x1, s_box[3] = y1, s_box[4] = z0, s_box[5] = z1;
            s_mask = mk;
            s_n = 0u;
        }
    }
    __syncthreads();
    const float x0 = s_box[0], y0 = s_box[1], x1 = s_box[2], y1 = s_box[3], z0 = s_box[4], z1 = s_box[5];
    const uint32_t mk = s_mask;
    int32_t *const list = const_cast<int32_t *>(t.near_list) + size_t(sb) * size_t(t.near_cap);
    for (int e0 = 0; e0 < n_eval; e0 += 256) { // block-uniform
        const int e = e0 + int(threadIdx.x);
        bool hit = false;
        if (e < n_eval) {
            const float4 f = t.p_txf[e];
            const uint32_t ch = uint32_t(t.p_ch[e]) & 31u;
            const float dx = fmaxf(fmaxf(x0 - f.x, f.x - x1), 0.f);
            const float dy = fmaxf(fmaxf(y0 - f.y, f.y - y1), 0.f);
            const float dz = fmaxf(fmaxf(z0 - f.z, f.z - z1), 0.f);
            hit = dist2_f32(dx, dy, dz) <= f.w && ((mk >> ch) & 1u) != 0u; // (the workgroups' own test, against the larger box)
        }
        const uint64_t hm = ballot64(hit);
        if (hm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_n, uint32_t(__popcll(hm)));
            base = uniform_u(base);
            if (hit) list[base + lane_prefix(hm)] = e;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) const_cast<uint32_t *>(t.near_cnt)[sb] = s_n;
    // (Padding the list to the next 1024 entries with -1, so that phase A can ask for its entries before it knows the length --
    // one round trip less per workgroup and tick -- was measured: 255 instead of 220 us per 64 ticks at a million receivers,
    // the three extra loads per thread cost more than the round trip.)
}

// A rank's frame list (one workgroup per tick of a batch of GATHERED ticks over a receiver partition).  The all-gather hands
// every rank the source indices of all ranks' frames; a region of the plane hears nothing of most of them, and what every
// later stage did per frame -- the pre-pass, the near-frame lists, the per-frame scans, the reorder stage's walk, the
// interference stages' index -- it did for all of them: a rank's time did not follow its links.  Here the tick's frames are
// tested ONCE against the union of the partition's filter-workgroup boxes with the workgroups' own expression (so a frame
// that could pass any workgroup's test passes this one: monotone in every |d|, the union box contains every box) and the
// survivors are compacted IN ORDER: the tick goes on as a tick of those frames only, its descriptor patched here on the
// device (n_active, n_cnt) -- the host sizes grids for all frames and never learns the count.  This launch is also the
// listed frames' pre-pass (k_tick_prep_batch is not launched for such a batch): their on-air records, their pre-filter
// records at the SWEEP's candidate level (sweep_level: a batch of overlapping SINR ticks sweeps at the sensitivity and lists
// at the interference floor), the counters later kernels add to.  Packets keep their global numbers: fl_map (local ->
// gathered slot) for the records' packet column, fl_lb (gathered slot -> listed frames before it) for the offsets by global
// number, and the per-packet Tx-failure flag of EVERY gathered slot is written here (it depends on the source's
// txProbability alone).  A frame whose source is one of the partition's own receivers is always kept (half duplex asks for
// no reach).  The ranks' node-table digests ride in the gathered blocks: a rank that built its records from another table
// than this one flags every tick (RM_ERR_STATE).
// (256 threads per tick, eight frames per thread and round -- 2048 frames per round, their loads all issued before any is used:
// a tick of a thousand frames is one round of independent round trips.  A workgroup of 1024 threads per tick was measured: alone
// on the device 36 us per 512 ticks, but 243 us with two other contexts' kernels in flight -- it waits for a compute unit with
// room for all sixteen of its waves.)
constexpr int kRfPer = 8;
__global__ void __launch_bounds__(kRfThreads) k_rank_frames(const NodesDev nd, const ModelDev m, TickDev *__restrict__ ticks, const RankFramesArgs a)
{
    constexpr int kWaves = kRfThreads / 64;
    __shared__ float s_box[kWaves][6];
    __shared__ uint32_t s_msk[kWaves], s_wcnt[kRfPer][kWaves];
    TickDev &t = ticks[blockIdx.x];
    const int tid = int(threadIdx.x), lane = tid & 63, wave = wave_index();
    const int T = t.n_pub;
    if (a.digest_off >= 0)
        for (int r = tid; r < a.world; r += kRfThreads) {
            const int32_t *d = a.gather_base + size_t(r) * size_t(a.gather_block) + size_t(a.digest_off);
            const uint64_t v = uint64_t(uint32_t(d[0])) | (uint64_t(uint32_t(d[1])) << 32);
            if (v != a.mine) t.stage_count[6] = 3u; // read as RM_ERR_STATE with the tick's result
        }
    if (T <= 0 || t.gather_idx == nullptr) return; // block-uniform: no list for this tick (the digests were all there was to do)
    // ---- the first round's source indices: requested before anything else
    int src[kRfPer];
    auto request = [&](const int i0) {
#pragma unroll
        for (int u = 0; u < kRfPer; ++u) {
            const int i = i0 + u * kRfThreads + tid;
            src[u] = -1;
            if (i < T) src[u] = t.gather_idx[size_t(i / t.gather_slots) * size_t(t.gather_stride) + size_t(i % t.gather_slots)];
        }
    };
    request(0);
    // ---- k_tick_prep's duties for the counters
    if (tid < 8) t.next_counters[tid] = 0u;
    for (int i = tid; i < kShards; i += kRfThreads) t.next_shard_count[i * kShardStride] = 0u;
    if (!t.use_matrix)
        for (int i = tid; i < t.zero_len; i += kRfThreads) {
            t.cursor[i] = 0u;
            t.cand_tot_next[i] = 0u;
        }
    if (t.reset_heads)
        for (int i = tid; i < t.n_rx; i += kRfThreads) t.head[i] = -1;
    if (t.acc_lo)
        for (int i = tid; i < t.n_rx; i += kRfThreads) {
            t.acc_lo[i] = 0ull;
            t.acc_hi[i] = 0ull;
        }
    // ---- the partition's box: the union of its filter workgroups' boxes and channel masks
    const float inf_ = __builtin_inff();
    float x0 = inf_, y0 = inf_, z0 = inf_, x1 = -inf_, y1 = -inf_, z1 = -inf_;
    uint32_t mk = 0u;
    const int n_wg = (t.n_rx + kGroup * 16 - 1) / (kGroup * 16);
    for (int w = tid; w < n_wg; w += kRfThreads) {
        const float4 q = nd.wg_box_xy[w];
        const float2 qz = nd.wg_box_z[w];
        x0 = fminf(x0, q.x), y0 = fminf(y0, q.y), x1 = fmaxf(x1, q.z), y1 = fmaxf(y1, q.w);
        z0 = fminf(z0, qz.x), z1 = fmaxf(z1, qz.y);
        mk |= nd.wg_chmask[w];
    }
    x0 = wave_min(x0), y0 = wave_min(y0), z0 = wave_min(z0);
    x1 = wave_max(x1), y1 = wave_max(y1), z1 = wave_max(z1);
    for (int d = 32; d >= 1; d >>= 1) mk |= uint32_t(__shfl_xor(int(mk), d));
    if (lane == 0) {
        s_box[wave][0] = x0, s_box[wave][1] = y0, s_box[wave][2] = z0, s_box[wave][3] = x1, s_box[wave][4] = y1, s_box[wave][5] = z1;
        s_msk[wave] = mk;
    }
    __syncthreads();
    mk = 0u;
    for (int w = 0; w < kWaves; ++w) {
        x0 = fminf(x0, s_box[w][0]), y0 = fminf(y0, s_box[w][1]), z0 = fminf(z0, s_box[w][2]);
        x1 = fmaxf(x1, s_box[w][3]), y1 = fmaxf(y1, s_box[w][4]), z1 = fmaxf(z1, s_box[w][5]);
        mk |= s_msk[w];
    }
    if (a.ring != nullptr) { // (block-uniform) frames that stay on the air: is every receiver still where the frames on the air were selected for?
        if (tid < kCullRing && tid != a.ring_slot) {
            const CullEntry e = a.ring[tid];
            if (e.end_us > a.t_first && (x0 < e.lo[0] || y0 < e.lo[1] || z0 < e.lo[2] || x1 > e.hi[0] || y1 > e.hi[1] || z1 > e.hi[2]))
                t.stage_count[6] = 4u; // read as RM_ERR_STATE with the tick's result
        }
        if (blockIdx.x == 0 && tid == 0) {
            CullEntry e;
            e.lo[0] = x0 - a.margin, e.lo[1] = y0 - a.margin, e.lo[2] = z0 - a.margin;
            e.hi[0] = x1 + a.margin, e.hi[1] = y1 + a.margin, e.hi[2] = z1 + a.margin;
            e.end_us = a.batch_end;
            a.ring[a.ring_slot] = e;
        }
    }
    x0 -= a.margin, y0 -= a.margin, z0 -= a.margin, x1 += a.margin, y1 += a.margin, z1 += a.margin;
    if (!a.use_chmask) mk = 0xFFFFFFFFu;
    const bool draws_possible = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_N2N || m.kind == RM_MODEL_LOGDIST);
    uint32_t base = 0; // frames listed so far (block-uniform)
    for (int i0 = 0; i0 < T; i0 += kRfPer * kRfThreads) { // block-uniform: one round per 2048 frames
        if (i0) request(i0);
        // ---- which frames matter here
        uint64_t hms[kRfPer];
        float4 fl[kRfPer]; // pre-filter records at the LIST's level
#pragma unroll
        for (int u = 0; u < kRfPer; ++u) {
            const int i = i0 + u * kRfThreads + tid;
            bool hit = false;
            fl[u] = make_float4(0.f, 0.f, 0.f, -1.f);
            if (i < T) {
                const rm_tx_record r = make_tx_record(nd, src[u], t.src_start_us, t.src_air_us);
                t.pkt_interference[i] = (draws_possible && tx_success(m, r) <= 0.0) ? 1 : 0; // (write_pkt_interference's rule, by global number)
                if (r.src >= 0) {
                    double thr64;
                    tx_prefilter(m, r, fl[u], thr64);
                    const float4 f = fl[u];
                    const float dx = fmaxf(fmaxf(x0 - f.x, f.x - x1), 0.f);
                    const float dy = fmaxf(fmaxf(y0 - f.y, f.y - y1), 0.f);
                    const float dz = fmaxf(fmaxf(z0 - f.z, f.z - z1), 0.f);
                    hit = dist2_f32(dx, dy, dz) <= f.w && ((mk >> (uint32_t(r.channel) & 31u)) & 1u) != 0u;
                    if (!hit) hit = engine_pos(nd, r.src) >= 0; // a receiver of this partition that is on the air itself: half duplex
                }
            }
            hms[u] = ballot64(hit);
            if (lane == 0) s_wcnt[u][wave] = uint32_t(__popcll(hms[u]));
        }
        __syncthreads();
        // ---- ordered compaction: frame i = i0 + u * 256 + tid comes after the frames of the chunks before u and of the waves before its own
#pragma unroll
        for (int u = 0; u < kRfPer; ++u) {
            const int i = i0 + u * kRfThreads + tid;
            uint32_t k = base + lane_prefix(hms[u]);
            uint32_t tot = 0;
            for (int w = 0; w < kWaves; ++w) {
                const uint32_t c = s_wcnt[u][w];
                if (w < wave) k += c;
                tot += c;
            }
            base += tot;
            if (i < T) t.fl_lb[i] = k;
            if (i < T && ((hms[u] >> lane) & 1ull)) {
                // (the listed frames' records are fetched again -- a fifth of the frames, L2-resident -- rather than kept for all eight)
                const rm_tx_record r = make_tx_record(nd, src[u], t.src_start_us, t.src_air_us);
                t.fl_map[k] = i;
                t.tx_build[k] = r;
                float4 f = fl[u];
                if (a.sweep_level != m.ld_level) { // (the sweep's own cut-off: the medium without SINR sweeps at the sensitivity)
                    double thr64;
                    tx_prefilter_at(m, a.sweep_level, r, f, thr64);
                }
                t.p_txf[k] = f;
                t.p_ch[k] = r.channel;
                t.p_src[k] = r.src;
                t.p_inv[k] = prefilter_inv(m, f);
            }
        }
        __syncthreads(); // (the next round's counts overwrite s_wcnt)
    }
    // (records kept on the air: the slots behind the listed frames hold padding, as the unlisted frames' own slots would
    // have -- whoever walks the window later finds records everywhere)
    if (t.fl_pad)
        for (int e = int(base) + tid; e < T; e += kRfThreads) t.tx_build[e] = make_tx_record(nd, -1, t.src_start_us, 0);
    if (tid == 0) {
        t.fl_lb[T] = base;
        t.n_active = int(base);
        t.n_cnt = max(kTxChunk, int((base + uint32_t(kTxChunk) - 1u) / uint32_t(kTxChunk)) * kTxChunk);
        t.gather_idx = nullptr; // (the records are in place: nothing of the gathered layout is needed any more)
        if (t.fl_ov_n_new) *t.fl_ov_n_new = int(base);
    }
}

// a lone tick over a window of frames that were selected for this partition (CullEntry): the same comparison, on its own
__global__ void __launch_bounds__(256) k_cull_check(const NodesDev nd, const CullEntry *__restrict__ ring, int64_t t_begin, uint32_t *flag_word)
{
    __shared__ float s_box[4][6];
    const int tid = int(threadIdx.x), lane = tid & 63, wave = wave_index();
    const float inf_ = __builtin_inff();
    float x0 = inf_, y0 = inf_, z0 = inf_, x1 = -inf_, y1 = -inf_, z1 = -inf_;
    const int n_wg = (nd.n_rx + kGroup * 16 - 1) / (kGroup * 16);
    for (int w = tid; w < n_wg; w += 256) {
        const float4 q = nd.wg_box_xy[w];
        const float2 qz = nd.wg_box_z[w];
        x0 = fminf(x0, q.x), y0 = fminf(y0, q.y), x1 = fmaxf(x1, q.z), y1 = fmaxf(y1, q.w);
        z0 = fminf(z0, qz.x), z1 = fmaxf(z1, qz.y);
    }
    x0 = wave_min(x0), y0 = wave_min(y0), z0 = wave_min(z0);
    x1 = wave_max(x1), y1 = wave_max(y1), z1 = wave_max(z1);
    if (lane == 0) s_box[wave][0] = x0, s_box[wave][1] = y0, s_box[wave][2] = z0, s_box[wave][3] = x1, s_box[wave][4] = y1, s_box[wave][5] = z1;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        x0 = fminf(x0, s_box[w][0]), y0 = fminf(y0, s_box[w][1]), z0 = fminf(z0, s_box[w][2]);
        x1 = fmaxf(x1, s_box[w][3]), y1 = fmaxf(y1, s_box[w][4]), z1 = fmaxf(z1, s_box[w][5]);
    }
    if (tid < kCullRing) {
        const CullEntry e = ring[tid];
        if (e.end_us > t_begin && (x0 < e.lo[0] || y0 < e.lo[1] || z0 < e.lo[2] || x1 > e.hi[0] || y1 > e.hi[1] || z1 > e.hi[2])) *flag_word = 4u;
    }
}

hipError_t launch_cull_check(hipStream_t s, const NodesDev &nd, const CullEntry *ring, int64_t t_begin, uint32_t *flag_word)
{
    RM_KLAUNCH(k_cull_check, dim3(1), dim3(256), 0, s, nd, ring, t_begin, flag_word);
    return hipGetLastError();
}

// a rank's block of a sharded batch as it goes into the all-gather: its source indices, then the trailer (the node table's
// digest in two words, two spare words)
__global__ void __launch_bounds__(256) k_stage_block(const int32_t *__restrict__ src, int n, uint64_t digest, int32_t *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
    if (i == 0) {
        dst[n] = int32_t(uint32_t(digest));
        dst[n + 1] = int32_t(uint32_t(digest >> 32));
        dst[n + 2] = 0;
        dst[n + 3] = 0;
    }
}

hipError_t launch_stage_block(hipStream_t s, const int32_t *src, int n, uint64_t digest, int32_t *dst)
{
    RM_KLAUNCH(k_stage_block, dim3(cdiv(max(n, 1), 256)), dim3(256), 0, s, src, n, digest, dst);
    return hipGetLastError();
}

hipError_t launch_rank_frames(hipStream_t s, const NodesDev &nd, const ModelDev &m, TickDev *dev_ticks, int n, int max_frames, const RankFramesArgs &a)
{
    if (n <= 0) return hipSuccess;
    (void)max_frames;
    RM_KLAUNCH(k_rank_frames, dim3(n), dim3(kRfThreads), 0, s, nd, m, dev_ticks, a);
    return hipGetLastError();
}

template <int RPT, bool SHADOW>
__global__ void __launch_bounds__(kBlock, 6)
k_filter_wg_batch(const NodesDev nd, const ModelDev m, const TickDev *__restrict__ ticks, const int n_ticks, const int per_wg)
{
    const int first = int(blockIdx.z) * per_wg;
    filter_wg_body<RPT, SHADOW>(nd, m, ticks, first, min(per_wg, n_ticks - first));
}

// The batch filter over the near-frame lists, SIXTEEN ticks at a time (RPT == 4: the large tables the lists are made for).
// With the lists a workgroup's phase A has a few dozen frames per tick to look at -- a few lanes of one round -- and what a
// tick cost was what every tick costs whatever its size: three dependent round trips (list length -> list entry -> record),
// three workgroup barriers and some eighty wave instructions of loop and bookkeeping per wave, 977 workgroups x 128 ticks of
// them per launch at a million receivers (two thirds of the filter's time there).  Here a wave asks for the lists of FOUR ticks
// per round (one per unrolled request: its lanes are the list's entries) and the workgroup's four waves for sixteen; the near
// frames of all of them are compacted into the same LDS list with their tick beside them, and phase B runs over full chunks of
// 64 near frames of mixed ticks: a frame's candidates go to ITS tick's shards (one allocation per frame and wave, by the lane
// the frame sits in), through its tick's descriptor.  The tests are filter_wg_tick's, expression for expression; the order of
// the candidates inside a tick's shards never mattered (the reorder stage sorts).
constexpr int kGrpTicks = 4 * kWavesPerBlock;

template <bool SHADOW>
__global__ void __launch_bounds__(kBlock, SHADOW ? 5 : 6)
k_filter_wg_group(const NodesDev nd, const ModelDev m, const TickDev *__restrict__ ticks, const int n_ticks, const int per_wg)
{
    constexpr int RPT = 4;
    __shared__ float4 s_txf[kNearLds];
    __shared__ int s_ch[kNearLds];
    __shared__ int s_e[kNearLds];
    __shared__ uint16_t s_tk[kNearLds]; // the frame's tick (index into `ticks`: at most RM_MAX_BATCH)
    __shared__ float s_inv[SHADOW ? kNearLds : 1];
    __shared__ int s_src[SHADOW ? kNearLds : 1];
    __shared__ uint64_t s_mask[kWavesPerBlock][kTxChunk][RPT];
    __shared__ uint32_t s_tbl[SHADOW ? kShadowBins : 1];
    __shared__ uint32_t s_n;

    const int first = int(blockIdx.z) * per_wg, count = min(per_wg, n_ticks - first);
    const int lane = threadIdx.x & 63;
    const int wave = wave_index();
    const int wg = blockIdx.x;
    const int sb = wg / kNearSb;
    const int slab = wg * kWavesPerBlock + wave;
    const int jbase = slab * (kGroup * RPT);
    WgRx<RPT> rx;
    wg_rx_load<RPT, SHADOW>(nd, ticks[first], rx); // (the receiver tiling is the same for every tick of a launch)
    const bool live = slab < ticks[first].n_slabs;
    const int n_rx = ticks[first].n_rx;
    const float (&fx)[RPT] = rx.fx, (&fy)[RPT] = rx.fy, (&fz)[RPT] = rx.fz;
    const int (&fch)[RPT] = rx.fch, (&forig)[RPT] = rx.forig;
    const float4 (&bxy)[RPT] = rx.bxy;
    const float2 (&bz)[RPT] = rx.bz;
    const float4 wxy = rx.wxy;
    const float2 wz = rx.wz;
    const uint32_t (&bmask)[RPT] = rx.bmask;
    const uint32_t wmask = rx.wmask;

    if (SHADOW) s_tbl[threadIdx.x] = m.shadow_tbl[threadIdx.x];
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();

    uint32_t round = 0;
    for (int b0 = 0; b0 < count; b0 += kGrpTicks) { // block-uniform
        const int g = min(kGrpTicks, count - b0);
        // the lists' lengths (every wave reads all sixteen: nothing to agree on through LDS)
        int cnt_l = 0;
        if (lane < g) {
            const TickDev &T = ticks[first + b0 + lane];
            cnt_l = int(min(T.near_cnt[sb], uint32_t(T.near_cap)));
        }
        const int rounds = (uniform_i(wave_max_i(cnt_l)) + 63) >> 6;
        constexpr int kUnrollA = 4;
        float4 af[kUnrollA];
        int ach[kUnrollA], asrc[kUnrollA], ae[kUnrollA];
        float ainv[kUnrollA];
        for (int r = 0; r < rounds; ++r) { // block-uniform
#pragma unroll
            for (int u = 0; u < kUnrollA; ++u) { // this wave's tick of request u: u * 4 + wave
                const int tb = u * kWavesPerBlock + wave;
                const int c = uniform_i(__shfl(cnt_l, tb));
                const int k = (r << 6) + lane;
                af[u] = make_float4(0.f, 0.f, 0.f, -1.f);
                ach[u] = 0;
                asrc[u] = -1;
                ainv[u] = 0.f;
                ae[u] = -1;
                if (k < c) {
                    const TickDev &T = ticks[first + b0 + tb];
                    const int e = T.near_list[size_t(sb) * size_t(T.near_cap) + size_t(k)];
                    ae[u] = e;
                    af[u] = T.p_txf[e];
                    ach[u] = T.p_ch[e];
                    if (SHADOW) {
                        ainv[u] = T.p_inv[e];
                        asrc[u] = T.p_src[e];
                    }
                }
            }
#pragma unroll 1
            for (int u = 0; u < kUnrollA; ++u) { // rolled: one copy of phase B; the records are selected, not indexed
                if (u * kWavesPerBlock >= g) break; // block-uniform
                float4 tfa = af[0];
                int cha = ach[0], srca = asrc[0], e = ae[0];
                float inva = ainv[0];
#pragma unroll
                for (int k = 1; k < kUnrollA; ++k) {
                    tfa.x = (u == k) ? af[k].x : tfa.x;
                    tfa.y = (u == k) ? af[k].y : tfa.y;
                    tfa.z = (u == k) ? af[k].z : tfa.z;
                    tfa.w = (u == k) ? af[k].w : tfa.w;
                    cha = (u == k) ? ach[k] : cha;
                    srca = (u == k) ? asrc[k] : srca;
                    inva = (u == k) ? ainv[k] : inva;
                    e = (u == k) ? ae[k] : e;
                }
                // phase A: this thread's frame against the workgroup box
                bool hit = false;
                if (e >= 0) {
                    const float dx = fmaxf(fmaxf(wxy.x - tfa.x, tfa.x - wxy.z), 0.f);
                    const float dy = fmaxf(fmaxf(wxy.y - tfa.y, tfa.y - wxy.w), 0.f);
                    const float dz = fmaxf(fmaxf(wz.x - tfa.z, tfa.z - wz.y), 0.f);
                    hit = dist2_f32(dx, dy, dz) <= tfa.w && ((wmask >> (uint32_t(cha) & 31u)) & 1u) != 0u; // (nobody here listens on its channel)
                }
                const uint64_t hm = ballot64(hit);
                if (hm) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_n, uint32_t(__popcll(hm)));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (hit) {
                        const uint32_t k = base + lane_prefix(hm);
                        s_txf[k] = tfa;
                        s_ch[k] = cha;
                        s_e[k] = e;
                        s_tk[k] = uint16_t(first + b0 + u * kWavesPerBlock + wave);
                        if (SHADOW) {
                            s_inv[k] = inva;
                            s_src[k] = srca;
                        }
                    }
                }
                __syncthreads();
                const int n_near = uniform_i(int(s_n));
                const bool last = r + 1 >= rounds && (u + 1 >= kUnrollA || (u + 1) * kWavesPerBlock >= g);
                if (!last && n_near + kBlock <= kNearLds) continue; // room for another 256 frames

                // phase B: chunks of 64 near frames (of any of the group's ticks), every wave for its own groups
                if (live) {
                    for (int c0 = 0; c0 < n_near; c0 += kTxChunk) {
                        const int nt = min(kTxChunk, n_near - c0);
                        uint64_t near[RPT];
                        uint64_t todo = 0;
                        {
                            const float4 tf = s_txf[c0 + min(lane, nt - 1)];
                            const uint32_t tchb = uint32_t(s_ch[c0 + min(lane, nt - 1)]) & 31u;
#pragma unroll
                            for (int q = 0; q < RPT; ++q) {
                                near[q] = 0;
                                if ((slab * RPT + q) * kGroup < n_rx) {
                                    const float dx = fmaxf(fmaxf(bxy[q].x - tf.x, tf.x - bxy[q].z), 0.f);
                                    const float dy = fmaxf(fmaxf(bxy[q].y - tf.y, tf.y - bxy[q].w), 0.f);
                                    const float dz = fmaxf(fmaxf(bz[q].x - tf.z, tf.z - bz[q].y), 0.f);
                                    near[q] = ballot64(lane < nt && dist2_f32(dx, dy, dz) <= tf.w && ((bmask[q] >> tchb) & 1u) != 0u);
                                }
                                todo |= near[q];
                            }
                        }
                        uint32_t my_total = 0;
                        uint64_t walk = todo;
                        while (walk) {
                            const int ti = __ffsll((long long)walk) - 1; // wave-uniform
                            walk &= walk - 1;
                            const float4 tf = s_txf[c0 + ti];
                            const int tch = s_ch[c0 + ti];
                            uint64_t mask[RPT];
                            uint32_t total = 0;
#pragma unroll
                            for (int q = 0; q < RPT; ++q) {
                                mask[q] = 0;
                                if ((near[q] >> ti) & 1ull) {
                                    const float s2 = dist2_f32(fx[q] - tf.x, fy[q] - tf.y, fz[q] - tf.z);
                                    bool h = (s2 <= tf.w) && (fch[q] == tch);
                                    if (SHADOW && h) {
                                        const int bin = min(kShadowBins - 1, int(s2 * s_inv[c0 + ti]));
                                        const uint32_t a = uint32_t(s_src[c0 + ti]), bb = uint32_t(forig[q]);
                                        const uint64_t key = (uint64_t(a < bb ? a : bb) << 32) | uint64_t(a < bb ? bb : a);
                                        h = uint32_t(mix64(m.ld_seed_mixed ^ key) >> 32) <= s_tbl[bin];
                                    }
                                    mask[q] = ballot64(h);
                                    total += uint32_t(__popcll(mask[q]));
                                }
                            }
                            if (total) {
                                if (lane == ti) my_total = total;
                                if (lane < RPT) {
                                    uint64_t v = mask[0];
#pragma unroll
                                    for (int q = 1; q < RPT; ++q) v = (lane == q) ? mask[q] : v;
                                    s_mask[wave][ti][lane] = v;
                                }
                            }
                        }
                        uint64_t have = ballot64(my_total != 0u);
                        if (have == 0) continue;
                        // the lane of a frame with candidates: its tick's counters and shards
                        uint32_t my_base = 0;
                        if (my_total != 0u) {
                            const int my_e = s_e[c0 + lane];
                            const int my_tk = int(s_tk[c0 + lane]);
                            const TickDev &T = ticks[my_tk];
                            if (!T.use_matrix && T.first_eval + my_e >= T.first_new) atomicAdd(&T.cand_tot[my_e - T.cnt_base], my_total);
                            const uint32_t shard = (uint32_t(slab) + (round + uint32_t(c0 >> 6)) * 37u + uint32_t(my_tk) * 101u + uint32_t(lane) * 7u) & T.shard_mask;
                            const uint32_t base = atomicAdd(&T.shard_count[shard * kShardStride], my_total);
                            if (base + my_total > T.seg_cap) { // the shard is full: drop the run, flag the tick
                                T.stage_count[1] = 1u;
                                my_total = 0u;
                            }
                            my_base = shard * T.seg_cap + base;
                        }
                        have = ballot64(my_total != 0u);
                        walk = have;
                        while (walk) {
                            const int ti = __ffsll((long long)walk) - 1;
                            walk &= walk - 1;
                            const uint32_t fbase = uniform_u(uint32_t(__shfl(int(my_base), ti)));
                            const int e_ti = s_e[c0 + ti];
                            const TickDev &T = ticks[uniform_i(int(s_tk[c0 + ti]))];
                            uint32_t pre = 0;
#pragma unroll
                            for (int q = 0; q < RPT; ++q) {
                                const uint64_t mk = s_mask[wave][ti][q];
                                if (mk == 0) continue;
                                if ((mk >> lane) & 1ull) {
                                    const uint32_t idx = fbase + pre + lane_prefix(mk);
                                    T.st_pkt[idx] = e_ti;
                                    T.st_dst[idx] = jbase + q * kGroup + lane;
                                    if (T.use_matrix) T.st_blk[idx] = fbase; // the run's base: only the ordered scatter of unsorted tables ranks inside it
                                }
                                pre += uint32_t(__popcll(mk));
                            }
                        }
                    }
                }
                round += uint32_t(kNearLds / kTxChunk);
                __syncthreads(); // every wave is done with the LDS records
                if (threadIdx.x == 0) s_n = 0u;
                __syncthreads();
                if (last) break;
            }
        }
    }
}

// ============================================================================ launchers

hipError_t launch_patch_nodes(hipStream_t s, const NodesDev &nd, const NodePatch *dev_list, int n, const NodePatch &one)
{
    if (n <= 0) return hipSuccess;
    RM_KLAUNCH(k_patch_nodes, dim3(cdiv(n, 256)), dim3(256), 0, s, nd, dev_list, n, one);
    return hipGetLastError();
}

hipError_t launch_prep_rx(hipStream_t s, const NodesDev &nd, const ModelDev &m)
{
    if (nd.n_rx <= 0) return hipSuccess;
    RM_KLAUNCH(k_prep_rx, dim3(cdiv(nd.n_rx, kGroup)), dim3(64), 0, s, nd, m);
    const int n_wg = cdiv(nd.n_rx, kGroup * 16);
    RM_KLAUNCH(k_wg_boxes, dim3(cdiv(n_wg, 256)), dim3(256), 0, s, nd, n_wg);
    return hipGetLastError();
}

hipError_t launch_pack_tx(hipStream_t s, const NodesDev &nd, const int32_t *dev_src, int n, int64_t start_us,
                          int64_t air_us, rm_tx_record *out)
{
    if (n <= 0) return hipSuccess;
    RM_KLAUNCH(k_pack_tx, dim3(cdiv(n, 256)), dim3(256), 0, s, nd, dev_src, n, start_us, air_us, out);
    return hipGetLastError();
}

hipError_t launch_pack_tx_batch(hipStream_t s, const NodesDev &nd, const int32_t *dev_src, int n_ticks, int n,
                                const int64_t *start_us, int64_t air_us, rm_tx_record *out, int world)
{
    if (n <= 0 || n_ticks <= 0) return hipSuccess;
    if (n_ticks > kMaxBatch) return hipErrorInvalidValue;
    for (int b0 = 0; b0 < n_ticks; b0 += kPackChunk) {
        const int nb = min(kPackChunk, n_ticks - b0);
        PackStarts st{};
        for (int b = 0; b < nb; ++b) st.start_us[b] = start_us[b0 + b];
        RM_KLAUNCH(k_pack_tx_batch, dim3(cdiv(n, 256), nb, max(world, 1)), dim3(256), 0, s, nd, dev_src, n, st, air_us, out, b0, n_ticks);
    }
    return hipGetLastError();
}

// Chooses the filter variant for this tick and fixes the receiver tiling (t.rpt, t.n_slabs):
//  kFilterGrid: k_filter on a (slab, tile) grid -- the general variant (fp64 frame, unsorted tables);
//  kFilterWg:   k_tick_prep + k_filter_wg, two-level cull inside one workgroup per 4*rpt groups.
PlanKnobs read_plan_knobs()
{
    PlanKnobs k{0, 0, false, 0};
    if (const char *e = getenv("RM_FILTER")) {
        if (!strcmp(e, "grid")) k.filter = kFilterGrid + 1;
        else if (!strcmp(e, "wg")) k.filter = kFilterWg + 1;
    }
    if (const char *e = getenv("RM_WG_RPT")) k.wg_rpt = (atoi(e) == 4) ? 4 : (atoi(e) == 2 ? 2 : 1);
    k.no_shadow_table = getenv("RM_NO_SHADOW_TABLE") != nullptr;
    return k;
}

int plan_filter(TickDev &t, const LaunchCfg &cfg, bool want_wg, const PlanKnobs &knobs)
{
    const int n_eval = t.n_active - t.first_eval;
    const int n_chunks = cdiv(max(n_eval, 1), kTxChunk);
    const long waves4 = long(cdiv(t.n_rx, 256)) * n_chunks;
    // enough waves to fill 256 CUs x 4 SIMDs several times over, else one group per wave
    t.rpt = (waves4 >= 4096) ? 4 : 1;
    t.n_slabs = cdiv(t.n_rx, 64 * t.rpt);
    int mode = kFilterGrid;
    if (cfg.bbox && !cfg.f64_filter) {
        const long pairs = long((cdiv(t.n_slabs, kWavesPerBlock) + 7) / 8 * 8) * n_chunks;
        // beyond a few thousand (workgroup, tile) pairs the 2-D grid of k_filter is mostly short-lived
        // workgroups that find nothing (1 M nodes, or thousands of frames on the air)
        if (t.rpt == 4 && pairs > 8192 && t.n_slabs >= 4 * 256) mode = kFilterWg;
        if (knobs.filter) mode = knobs.filter - 1;
        if (want_wg) mode = kFilterWg;
        if (mode == kFilterWg) {
            // batches bring their own parallelism (workgroups x ticks): the coarse tiling halves the frame x
            // workgroup-box tests of phase A twice over; a lone tick needs the workgroups
            // (a receiver partition's few tiles per tick are enough when the batch has hundreds of ticks: 13 workgroups x 256)
            int rpt = (t.n_rx > 400000 || (want_wg && (t.n_rx >= 16384 || long(t.n_rx) * knobs.batch_ticks >= (1L << 20)))) ? 4 : 1;
            if (knobs.wg_rpt) rpt = knobs.wg_rpt;
            t.rpt = rpt;
            t.n_slabs = cdiv(t.n_rx, 64 * t.rpt);
        }
    }
    t.filter_mode = mode;
    return mode;
}

hipError_t launch_filter(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                         const LaunchCfg &cfg)
{
    const int n_eval = t.n_active - t.first_eval;
    if (n_eval <= 0 || t.n_slabs <= 0) return hipSuccess;
    if (t.filter_mode == kFilterWg) {
        RM_KLAUNCH(k_tick_prep, dim3(cdiv(n_eval, 256)), dim3(256), 0, s, nd, m, t);
        const dim3 grid(cdiv(t.n_slabs, kWavesPerBlock)), block(kBlock);
        if (t.rpt == 4) {
            if (cfg.shadow) RM_KLAUNCH((k_filter_wg<4, true>), grid, block, 0, s, nd, m, t);
            else RM_KLAUNCH((k_filter_wg<4, false>), grid, block, 0, s, nd, m, t);
        } else if (t.rpt == 2) {
            if (cfg.shadow) RM_KLAUNCH((k_filter_wg<2, true>), grid, block, 0, s, nd, m, t);
            else RM_KLAUNCH((k_filter_wg<2, false>), grid, block, 0, s, nd, m, t);
        } else {
            if (cfg.shadow) RM_KLAUNCH((k_filter_wg<1, true>), grid, block, 0, s, nd, m, t);
            else RM_KLAUNCH((k_filter_wg<1, false>), grid, block, 0, s, nd, m, t);
        }
        return hipGetLastError();
    }
    // XCD-aware launch: workgroups are dealt round-robin over the 8 XCDs in linear order (x fastest),
    // so with gridDim.x a multiple of 8 every tile-workgroup of one receiver slab has the same
    // blockIdx.x % 8 -- one XCD, one L2 -- and the slab's records leave HBM once per tick, not once
    // per XCD (placement is a speed matter only; the padding workgroups exit at once)
    const dim3 grid((cdiv(t.n_slabs, kWavesPerBlock) + 7) / 8 * 8, cdiv(n_eval, kTxChunk));
    const dim3 block(kBlock);
#define RM_LAUNCH(RPT, F64, BBOX, SH) RM_KLAUNCH((k_filter<RPT, F64, BBOX, SH>), grid, block, 0, s, nd, m, t)
    if (t.rpt == 4) {
        if (cfg.f64_filter) RM_LAUNCH(4, true, false, false);
        else if (cfg.bbox && cfg.shadow) RM_LAUNCH(4, false, true, true);
        else if (cfg.bbox) RM_LAUNCH(4, false, true, false);
        else if (cfg.shadow) RM_LAUNCH(4, false, false, true);
        else RM_LAUNCH(4, false, false, false);
    } else {
        if (cfg.f64_filter) RM_LAUNCH(1, true, false, false);
        else if (cfg.bbox && cfg.shadow) RM_LAUNCH(1, false, true, true);
        else if (cfg.bbox) RM_LAUNCH(1, false, true, false);
        else if (cfg.shadow) RM_LAUNCH(1, false, false, true);
        else RM_LAUNCH(1, false, false, false);
    }
#undef RM_LAUNCH
    return hipGetLastError();
}

// ticks of a batch a filter workgroup sweeps with ONE load of its receivers (rm_batch_tile_reuse reports it: the receiver
// table leaves HBM once per that many ticks of a launch)
int filter_ticks_per_wg(const TickDev &t0, int n)
{
    const int tiles = cdiv(t0.n_slabs, kWavesPerBlock);
    int per_wg = max(1, min(n, (tiles * n) / 3072));
    if (const char *e = getenv("RM_FILTER_TICKS_PER_WG")) per_wg = max(1, min(n, atoi(e)));
    return per_wg;
}

// rm_batch_*, stage 0: every tick's pre-pass and two-level filter (blockIdx.z = tick); `ticks` are the host
// copies of the descriptors (grid sizes), `b` the same descriptors in device memory
hipError_t launch_filter_batch(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n, const TickDev *b,
                               const LaunchCfg &cfg)
{
    int max_eval = 0;
    for (int i = 0; i < n; ++i) max_eval = max(max_eval, ticks[i].n_active - ticks[i].first_eval);
    const TickDev &t0 = ticks[0];
    if (t0.n_pub <= 0) // (a rank's frame lists: k_rank_frames was this batch's pre-pass as well)
        RM_KLAUNCH(k_tick_prep_batch, dim3(cdiv(max_eval, 256), 1, n), dim3(256), 0, s, nd, m, b);
    if (t0.near_list != nullptr) { // (every tick of the batch has its lists, or none has)
        const int n_wg = cdiv(t0.n_rx, kGroup * 16);
        RM_KLAUNCH(k_near_lists, dim3(cdiv(n_wg, kNearSb), 1, n), dim3(256), 0, s, nd, b, n_wg);
    }
    // A workgroup keeps its receivers for `per_wg` ticks: as many as leave a few thousand workgroups for the chip (a table
    // of a million receivers has a thousand tiles: 16 ticks = 4 per workgroup; 100 k receivers: one tick per workgroup).
    const int tiles = cdiv(t0.n_slabs, kWavesPerBlock);
    const int per_wg = filter_ticks_per_wg(t0, n);
    const dim3 grid(tiles, 1, cdiv(n, per_wg)), block(kBlock);
    // The lists of sixteen ticks at a time pay where a list is a few dozen frames -- many blocks (a million receivers), or few
    // frames for this partition's part of the plane; with lists of hundreds of frames (configs[2] / [3] at 100 k receivers: seven
    // blocks) a round of the per-tick form is already full and the grouped one was measured 6 % / 13 % slower.
    // RM_FILTER_GROUP=0 / 1: never / whenever the launch allows it (read per launch: tests).
    const char *e_grp = getenv("RM_FILTER_GROUP");
    const int n_sb = cdiv(tiles, kNearSb);
    const double seen = double(max_eval) * fmin(1.0, 1.3 * double(t0.n_rx) / double(max(nd.n, 1)) + 0.02); // frames this partition keeps
    bool group = n_sb >= 32 || seen * 1.5 / double(n_sb) <= 128.0;
    if (e_grp) group = atoi(e_grp) != 0;
    if (t0.rpt == 4 && t0.near_list != nullptr && per_wg > 1 && group) {
        if (cfg.shadow) RM_KLAUNCH((k_filter_wg_group<true>), grid, block, 0, s, nd, m, b, n, per_wg);
        else RM_KLAUNCH((k_filter_wg_group<false>), grid, block, 0, s, nd, m, b, n, per_wg);
    } else if (t0.rpt == 4) {
        if (cfg.shadow) RM_KLAUNCH((k_filter_wg_batch<4, true>), grid, block, 0, s, nd, m, b, n, per_wg);
        else RM_KLAUNCH((k_filter_wg_batch<4, false>), grid, block, 0, s, nd, m, b, n, per_wg);
    } else if (t0.rpt == 2) {
        if (cfg.shadow) RM_KLAUNCH((k_filter_wg_batch<2, true>), grid, block, 0, s, nd, m, b, n, per_wg);
        else RM_KLAUNCH((k_filter_wg_batch<2, false>), grid, block, 0, s, nd, m, b, n, per_wg);
    } else {
        if (cfg.shadow) RM_KLAUNCH((k_filter_wg_batch<1, true>), grid, block, 0, s, nd, m, b, n, per_wg);
        else RM_KLAUNCH((k_filter_wg_batch<1, false>), grid, block, 0, s, nd, m, b, n, per_wg);
    }
    return hipGetLastError();
}

} // namespace rm
