// rm_device.hpp -- device-side building blocks shared by the kernels: wave helpers, pre-filter records, eval_link, fused scans
// (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math; overview at the top of rm_engine.h)
#pragma once

#include "rm_math.hpp"

namespace rm {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

RM_D float wave_min(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d));
    return v;
}
RM_D float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}
RM_D int wave_max_i(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}

RM_D float round_up_to_float(double v)
{
    float f = float(v);
    if (double(f) < v) f = nextafterf(f, __builtin_inff());
    return f;
}

// Pre-filter record of one frame: position in the fp32 frame + threshold on the squared fp32
// distance (and the fp64 threshold for the fp64 variant).  thr < 0: nobody can be a candidate;
// thr = +inf: every enabled same-channel receiver is one (non-geometric media, or a frame whose
// position lies outside the frame the fp32 slack was computed for).
RM_D void tx_prefilter_at(const ModelDev &m, const double level, const rm_tx_record &tx, float4 &f, double &thr64);
RM_D void tx_prefilter(const ModelDev &m, const rm_tx_record &tx, float4 &f, double &thr64) { tx_prefilter_at(m, m.ld_level, tx, f, thr64); }
// (the same with the level given: the log-distance medium's cut-off at another level than the model's candidate level)
RM_D void tx_prefilter_at(const ModelDev &m, const double level, const rm_tx_record &tx, float4 &f, double &thr64)
{
    const double inf = u2f(0x7FF0000000000000ull);
    double cut; // cut-off distance (metres): no link beyond it can matter
    if (tx.src < 0) {
        cut = -1.0; // padding record
    } else if (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_UDGM_CONST) {
        cut = m.geo_cut;
    } else if (m.kind == RM_MODEL_LOGDIST) {
        const double margin = tx.txpower - m.ld_pl0 + m.ld_sigma * m.ld_clip - (level - 1e-6);
        if (!(margin >= 0.0)) {
            cut = -1.0;
        } else if (!(m.ld_exp > 0.0)) {
            cut = inf;
        } else {
            // hardware fp32 exp2 (relative error ~1e-6 at these arguments) with a 1e-4 pad
            cut = m.ld_d0 * double(__builtin_amdgcn_exp2f(float(margin * m.ld_cut_scale))) * (1.0 + 1e-4);
            if (cut < m.ld_d0) cut = m.ld_d0;
        }
    } else {
        cut = inf; // Null / N2N: no geometry
    }
    const double rx_ = tx.x - m.org_x, ry_ = tx.y - m.org_y, rz_ = tx.z - m.org_z;
    const bool geometric = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_UDGM_CONST || m.kind == RM_MODEL_LOGDIST);
    const bool in_frame = fabs(rx_) <= m.coord_bound && fabs(ry_) <= m.coord_bound && fabs(rz_) <= m.coord_bound;
    f.x = f.y = f.z = 0.f;
    if (cut < 0.0) {
        f.w = -1.f;
        thr64 = -1.0;
    } else if (!geometric || !in_frame || cut == inf) {
        f.w = __builtin_inff();
        thr64 = inf;
        if (geometric && in_frame) {
            f.x = float(rx_);
            f.y = float(ry_);
            f.z = float(rz_);
        }
    } else {
        f.x = float(rx_);
        f.y = float(ry_);
        f.z = float(rz_);
        const double eps = 0x1.0p-24;
        const double c = cut + m.f32_slack + 4.0 * eps * cut;
        f.w = round_up_to_float(c * c * (1.0 + 16.0 * eps));
        thr64 = (cut * cut) * (1.0 + 1e-12);
    }
}

// One frame on the air, indexed for the second launch of a tick by scan (rm_airscan.hip; ScanDev):
// where it is in the fp32 frame and how far it can matter at the level of `m` -- the pre-filter's own cut-off, as a radius --,
// its cell of the frame grid, and its place in its source node's chain of frames (half duplex does not ask for reach).
RM_D float sqrt_up(float v) { return nextafterf(__builtin_sqrtf(v), __builtin_inff()); }
RM_D int sg_cell1(float x, float half, float inv) // (monotone in x: a range of positions maps to a range of cells)
{
    const int c = int((x + half) * inv);
    return min(max(c, 0), kSgG - 1);
}
RM_D void scan_index(const ModelDev &m, const TickDev &t, const ScanDev &sd, int n_nodes, int i, const rm_tx_record &r)
{
    float4 f;
    double thr64;
    tx_prefilter(m, r, f, thr64);
    const bool on_air = r.src >= 0 && r.src < n_nodes && r.start_us + r.air_us > t.air.t_begin;
    float rad = -1.f;
    if (on_air && f.w >= 0.f) rad = (f.w < __builtin_inff()) ? sqrt_up(f.w) : __builtin_inff();
    sd.xyzr[i] = make_float4(f.x, f.y, f.z, rad);
    sd.ch[i] = r.channel;
    if (!on_air) return;
    const unsigned long long stamp = (unsigned long long)sd.stamp << 32;
    const unsigned long long old = atomicExch(&sd.self_slot[r.src], stamp | (unsigned long long)uint32_t(i));
    sd.self_next[i] = ((old >> 32) == sd.stamp) ? int(uint32_t(old)) : -1;
    if (rad < 0.f) return;
    bool placed = false;
    if (rad < __builtin_inff()) {
        const int cell = sg_cell1(f.y, sd.half, sd.inv) * kSgG + sg_cell1(f.x, sd.half, sd.inv);
        const uint32_t k = atomicAdd(&sd.cnt[cell], 1u);
        if (k < uint32_t(kSgK)) {
            sd.bucket_xyzr[cell * kSgK + int(k)] = make_float4(f.x, f.y, f.z, rad);
            sd.bucket_ci[cell * kSgK + int(k)] = make_int2(r.channel, i);
            placed = true;
        }
        atomicMax(&sd.cnt[kSgCells + 1 + (i & (kSgMax - 1))], __float_as_uint(rad)); // (radii are >= 0: their bits order like they do)
    }
    if (!placed) sd.every[atomicAdd(&sd.cnt[kSgCells], 1u)] = uint32_t(i); // no bound, or the cell is full: everybody looks at it
}

// engine position of a node in this partition's receiver table, -1 = not a receiver here (another rank's node, padding)
RM_D int engine_pos(const NodesDev &nd, int node)
{
    const uint32_t k = uint32_t(node - nd.rx_first);
    return (k < uint32_t(nd.pos_span)) ? nd.pos_of[k] : -1;
}

// RadioPacket(node, time, data): copies the source radio's txpower / channel (RadioPacket.java:46-52)
RM_D rm_tx_record make_tx_record(const NodesDev &nd, int s, int64_t start_us, int64_t air_us)
{
    rm_tx_record r;
    if (s < 0 || s >= nd.n) { // padding slot
        r.x = r.y = r.z = 0.0;
        r.txpower = 0.0;
        r.txprob = 0.0;
        r.start_us = start_us;
        r.air_us = 0;
        r.src = -1;
        r.channel = 0;
    } else {
        const SrcRecord sr = nd.srec[s]; // (one line: position, power, probability, channel)
        r.x = sr.x;
        r.y = sr.y;
        r.z = sr.z;
        r.txpower = sr.txpower;
        r.txprob = sr.txprob;
        r.start_us = start_us;
        r.air_us = air_us;
        r.src = s;
        r.channel = sr.channel;
    }
    return r;
}

// ---- wave-level helpers

RM_D uint32_t lane_prefix(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

RM_D uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Values that are the same in every lane of a wave but that the compiler cannot know to be (the wave
// index, anything read from LDS or memory at a wave-uniform address): moved to a scalar register,
// so that the loops and branches they steer run on the scalar unit instead of as masked vector code.
RM_D int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
RM_D uint32_t uniform_u(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
RM_D int wave_index() { return __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)); }

// Consecutive lanes with equal `key` form a run (candidate entries of one frame are contiguous in
// the list).  For the lanes with `pred`: how many such lanes precede me inside my run, how many
// the run has, and which lane leads it -- so that one atomic per run replaces one per link.
struct RunInfo {
    int start;
    uint32_t before, total;
};
RM_D RunInfo run_prefix(int key, bool pred, int lane)
{
    const int prev = __shfl_up(key, 1);
    const uint64_t starts = ballot64(lane == 0 || key != prev);
    const uint64_t preds = ballot64(pred);
    const uint64_t upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull); // lanes 0..lane
    RunInfo r;
    r.start = 63 - __clzll((long long)(starts & upto));
    const uint64_t later = starts & ~upto;
    const int end = later ? (__ffsll((long long)later) - 1) : 64;
    const uint64_t run = ((end == 64) ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << r.start) - 1ull);
    r.before = uint32_t(__popcll(preds & run & ((1ull << lane) - 1ull)));
    r.total = uint32_t(__popcll(preds & run));
    return r;
}

// squared fp32 distance; fma is fine here: the filter only has to be conservative, and the box
// test uses the very same expression (monotone in each |d|)
RM_D float dist2_f32(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

// ---- one link, the reference's way

struct LinkEval {
    bool append;   // stays in the link list
    bool wanted;   // heard link of a new frame: gets an output record
    uint8_t flags;
    double aux;    // probability (UDGM / N2N) or rssi (logdist)
    double lin;    // linear power (SINR)
};

// Exact evaluation of one link, in the reference's order of tests
// (UDGMRadioMedium.java:99-111, N2NRadioMedium.java:55-67, NullRadioMedium.java:62-73,
//  UDGMConstantLossRadioMedium.java:25-33).  `pos` is the receiver's engine position.
template <int MODEL, bool SINR>
RM_D LinkEval eval_link(const ModelDev &m, const NodesDev &nd, const rm_tx_record &tx, const RxRecord &rx_, bool is_new)
{
    LinkEval r;
    r.append = false;
    r.wanted = false;
    r.flags = 0;
    r.aux = 0.0;
    r.lin = 0.0;
    const int j = rx_.orig;
    if (j == tx.src) return r;                  // node != source
    if (!rx_.enabled) return r;                 // radio.isEnabled()
    if (rx_.channel != tx.channel) return r;    // radio.getWirelessChannel() == channel
    if (MODEL == RM_MODEL_NULL) {
        r.append = r.wanted = is_new;
        r.flags = kFlagHeardNew;
        return r;
    }
    if (MODEL == RM_MODEL_N2N) {
        // N2NRadioMedium.java:28-37
        const int sid = nd.sint_id[tx.src];
        const int did = rx_.int_id;
        double p = 0.0;
        if (m.n2n != nullptr && sid > 0 && did > 0 && sid <= m.n2n_m && did <= m.n2n_m) {
            p = m.n2n[int64_t(sid - 1) * m.n2n_m + (did - 1)] * rx_.rxprob;
        }
        if (p <= 0.0) return r;
        r.append = r.wanted = is_new;
        r.flags = kFlagHeardNew;
        r.aux = p;
        return r;
    }
    const double rx = rx_.x, ry = rx_.y, rz = rx_.z;
    if (MODEL == RM_MODEL_UDGM_CONST) {
        const double d = ref_distance(tx.x, tx.y, tx.z, rx, ry, rz);
        if (d < m.const_range) {
            r.append = r.wanted = is_new;
            r.flags = kFlagHeardNew;
        }
        return r;
    }
    if (MODEL == RM_MODEL_UDGM) {
        // UDGMRadioMedium.java:67-81 ; Math.pow(v, 2.0) == v*v
        const double d = ref_distance(tx.x, tx.y, tx.z, rx, ry, rz);
        const double d2 = d * d;
        const double dmax = m.udgm_range;
        if (dmax == 0.0) return r;
        const double dmax2 = dmax * dmax;
        double ratio = d2 / dmax2;
        if (ratio > 1.0) return r;
        ratio = 1.0 - ratio * (1.0 - m.udgm_ratio_rx);
        const double p = ratio * rx_.rxprob;
        if (p <= 0.0) return r;
        r.append = r.wanted = is_new;
        r.flags = kFlagHeardNew;
        r.aux = p;
        return r;
    }
    if (MODEL == RM_MODEL_LOGDIST) {
        const double rssi = logdist_rssi(m, tx, rx, ry, rz, j);
        const bool heard = is_new && (rssi >= m.ld_sens) && !(rx_.rxprob <= 0.0);
        r.aux = rssi;
        if (SINR) {
            const bool interferer = rssi >= m.ld_ifloor;
            if (!heard && !interferer) return r;
            r.append = true;
            r.wanted = heard;
            r.flags = uint8_t((heard ? kFlagHeardNew : 0) | (interferer ? kFlagInterferer : 0));
            if (interferer) r.lin = det_pow10(rssi / 10.0);
        } else if (heard) {
            r.append = r.wanted = true;
            r.flags = kFlagHeardNew;
        }
        return r;
    }
    return r;
}

// Exclusive scan of n per-frame counts inside one 256-thread workgroup, result in LDS (and in
// `pub` if not null).  Every workgroup of a consumer kernel redoes it (T counts, a few KB from L2)
// instead of paying a separate kernel for it.  Returns the total; *vmax gets the largest count.
RM_D uint32_t block_scan_counts(const uint32_t *cnt, int n, uint32_t *s_off, uint32_t *s_wave /*[4]*/, uint32_t *pub,
                                uint32_t *vmax_out)
{
    const int per = (n + 255) / 256;
    const int i0 = min(n, int(threadIdx.x) * per), i1 = min(n, i0 + per);
    uint32_t sum = 0, vmax = 0;
    for (int i = i0; i < i1; ++i) {
        const uint32_t v = cnt[i];
        sum += v;
        vmax = max(vmax, v);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    for (int i = i0; i < i1; ++i) {
        s_off[i] = run;
        if (pub) pub[i] = run;
        run += cnt[i];
    }
    if (pub && threadIdx.x == 0) pub[n] = total;
    if (threadIdx.x == 0) s_off[n] = total; // (a consumer takes a frame's count as the difference of two offsets)
    if (vmax_out) {
        for (int d = 32; d >= 1; d >>= 1) vmax = max(vmax, uint32_t(__shfl_xor(int(vmax), d)));
        *vmax_out = vmax;
    }
    __syncthreads();
    return total;
}

// The same for at most 256*PER counts (PER = 4: the bench's 1000 frames per tick, 16: up to 4096):
// PER counts per thread, requested with small_scan_load at the top of the kernel so that the
// round trip overlaps the kernel's own first loads, and only 1 KB * PER of LDS (the occupancy of
// the consumers is LDS-bound with the general 32 KB variant).
constexpr int kSmallScan = 1024, kMediumScan = 4096;
template <int PER> struct SmallCounts {
    uint32_t v[PER];
};
template <int PER> RM_D SmallCounts<PER> small_scan_load(const uint32_t *cnt, int n)
{
    SmallCounts<PER> c;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = int(threadIdx.x) * PER + k;
        c.v[k] = (i < n) ? cnt[i] : 0u;
    }
    return c;
}
template <int PER>
RM_D uint32_t small_scan(const SmallCounts<PER> &c, int n, uint32_t *s_off, uint32_t *s_wave /*[4]*/, uint32_t *pub, uint32_t *vmax_out)
{
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) sum += c.v[k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = int(threadIdx.x) * PER + k;
        if (i < n) {
            s_off[i] = run;
            if (pub) pub[i] = run;
        }
        run += c.v[k];
    }
    if (pub && threadIdx.x == 0) pub[n] = total;
    if (threadIdx.x == 0) s_off[n] = total;
    if (vmax_out) {
        uint32_t vmax = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) vmax = max(vmax, c.v[k]);
        for (int d = 32; d >= 1; d >>= 1) vmax = max(vmax, uint32_t(__shfl_xor(int(vmax), d)));
        // the publisher needs the maximum over the whole workgroup
        __shared__ uint32_t s_vmax[4];
        if (lane == 0) s_vmax[wave] = vmax;
        __syncthreads();
        *vmax_out = max(max(s_vmax[0], s_vmax[1]), max(s_vmax[2], s_vmax[3]));
    }
    __syncthreads();
    return total;
}
// scan variant of a kernel template parameter: 1 general (<= kFusedScanMax), 3 small, 4 medium
constexpr int scan_per(int v) { return v == 3 ? 4 : 16; }
constexpr int scan_lds(int v) { return v == 1 ? kFusedScanMax + 1 : (v == 3 ? kSmallScan + 1 : (v == 4 ? kMediumScan + 1 : 1)); }
static inline int scan_variant(int n_cnt) { return n_cnt <= kSmallScan ? 3 : (n_cnt <= kMediumScan ? 4 : (n_cnt <= kFusedScanMax ? 1 : 2)); }

RM_D double tx_success(const ModelDev &m, const rm_tx_record &tx)
{
    // UDGMRadioMedium.java:63-65 uses successRatioRx (sic); N2NRadioMedium.java:24-26
    if (m.kind == RM_MODEL_UDGM) return m.udgm_ratio_rx * tx.txprob;
    return tx.txprob;
}

// ---- 1024-thread scans

RM_D uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread over a 1024-thread block; returns the block total in `total`
RM_D uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *s_wave /*[16]*/, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t x = s_wave[w];
        if (w < wave) wave_off += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return wave_off + inc - v;
}

// per-packet Tx-failure flag where no draw can happen (txSuccess <= 0 is the only way to fail)
// ---- the on-air lists across ticks (AirDev, rm_engine.h)
// room for the wanted lanes' entries in sub-ring `sub`: one atomic per wave.  Whole waves call this together.
RM_D int air_alloc(const TickDev &t, bool want, uint32_t sub)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    const uint32_t cnt = uint32_t(__popcll(mask));
    uint32_t base = 0;
    if (lane == leader) {
        base = atomicAdd(&t.air.tail[sub * kShardStride], cnt);
        const uint32_t live0 = t.air.mark[(t.air.wtick & (kAirTicks - 1)) * kShards + sub];
        if (base + cnt - live0 > t.air.sub_mask + 1u) { // would overwrite entries of frames still on the air
            t.stage_count[1] = 1u;
            t.air.bad[0] = 1u; // sticky: every later tick is dropped as well until the host has rebuilt the lists
        }
    }
    base = uint32_t(__shfl(int(base), leader));
    if (!want) return -1;
    const uint32_t seq = base + uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
    return int((sub << t.air.sub_shift) | (seq & t.air.sub_mask));
}

RM_D uint32_t air_sub(const TickDev &t)
{
    return (blockIdx.x * 4u + (threadIdx.x >> 6) + blockIdx.y * 17u + t.air.tick * 61u) & uint32_t(kShards - 1);
}

// the entry becomes the head of its receiver's list; the old head is kept as `next` only if it is still live
RM_D void air_link(const TickDev &t, int aidx, int pos, int64_t start_us, int64_t air_us, double lin, uint32_t flags)
{
    const unsigned long long mine = ((unsigned long long)t.air.tick << 32) | uint32_t(aidx);
    const unsigned long long old = atomicExch(&t.air.head[pos], mine);
    AirEntry e;
    e.start_us = start_us;
    e.lin = lin;
    e.air_us = uint32_t(air_us);
    if ((unsigned long long)air_us >> 32) { // (records in device memory are not seen by the host: a frame of 2^32 us or more)
        t.stage_count[1] = 1u;
        t.air.bad[0] = 1u;
    }
    e.next = (uint32_t(old >> 32) >= t.air.wtick) ? int(uint32_t(old)) : -1; // an empty head has tick 0
    e.meta = (t.air.tick << 2) | flags;
    e.pad = 0u;
    t.air.pool[aidx] = e;
}

// The tick is over for the lists: the next tick's entries begin where the sub-rings' tails are now.  One workgroup of
// kShards threads, after the last allocation of the tick (a later kernel than the one that evaluates the links).
RM_D void air_end(const TickDev &t)
{
    if (threadIdx.x < uint32_t(kShards))
        t.air.mark[((t.air.tick + 1u) & (kAirTicks - 1)) * kShards + threadIdx.x] = t.air.tail[threadIdx.x * kShardStride];
}

struct SinrOut {
    double sinr;
    bool collided;
};

// SINR of one heard link of a new frame (start w_start, length w_air) at the receiver in engine position `pos`: the
// receiver's list holds every co-channel frame on the air that is significant there; the interferers that overlap the
// frame in time are summed exactly (Q80: the order of the list does not matter), a SELF entry is half duplex.
// `self` is the link's own entry -- or kAirOwnInSum when the caller does not know it: the link's own entry is then
// summed like every other and taken out again afterwards, which is exact in Q80 (the entry exists iff the link is an
// interferer itself, rssi >= ifloor, and it is counted iff the frame overlaps itself and has not left the air).
constexpr int kAirOwnInSum = -2;
RM_D SinrOut air_sinr(const ModelDev &m, const TickDev &t, int pos, int self, int64_t w_start, int64_t w_air, double rssi)
{
    U128 acc = {0, 0};
    bool half_duplex = false;
    const int64_t w_end = w_start + w_air;
    const unsigned long long h = t.air.head[pos];
    uint32_t prev = uint32_t(h >> 32);
    int idx = (prev >= t.air.wtick) ? int(uint32_t(h)) : -1;
    for (int hops = 0; idx >= 0 && hops < (1 << 22); ++hops) {
        const AirEntry k = t.air.pool[idx];
        const uint32_t kt = k.meta >> 2;
        if (kt > prev || kt < t.air.wtick) break; // a slot that was handed out again: the list ended before it
        prev = kt;
        const int64_t k_end = k.start_us + int64_t(k.air_us);
        if (idx != self && k_end > t.air.t_begin && k.start_us < w_end && k_end > w_start) {
            if (k.meta & kAirSelf) half_duplex = true;
            else acc = u128_add(acc, q80_from_double(k.lin));
        }
        idx = k.next;
    }
    if (self == kAirOwnInSum && rssi >= m.ld_ifloor && w_air > 0 && w_end > t.air.t_begin)
        acc = u128_sub(acc, q80_from_double(det_pow10(rssi / 10.0))); // eval_link's lin of this very link
    SinrOut r;
    r.sinr = rssi - 10.0 * det_log10(q80_to_double(acc) + m.ld_noise_lin);
    r.collided = half_duplex || !(r.sinr >= m.ld_capture);
    return r;
}

RM_D void write_pkt_interference(const ModelDev &m, const TickDev &t, uint32_t first, uint32_t stride)
{
    const bool draws_possible = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_N2N || m.kind == RM_MODEL_LOGDIST);
    const int n_new = t.n_active - t.first_new;
    for (uint32_t q = first; q < uint32_t(n_new); q += stride) {
        const rm_tx_record tx = t.tx[t.first_new + q];
        t.pkt_interference[q] = (draws_possible && tx_success(m, tx) <= 0.0) ? 1 : 0;
    }
}

// the smallest fused-scan variant that holds every tick's counts
static inline int batch_scan_variant(const TickDev *ticks, int n)
{
    int scan = 3;
    for (int i = 0; i < n; ++i) {
        const int v = scan_variant(ticks[i].n_cnt);
        scan = (v == 1 || scan == 1) ? 1 : max(scan, v);
    }
    return scan;
}

} // namespace rm
