// rm_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the radio-medium engine.
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

#include "rm_engine.h"

#include <stdlib.h>
#include <string.h>

#include <math.h>

namespace rm {

#define RM_HD __host__ __device__ inline
#define RM_D __device__ inline

// ============================================================================ exact math
// Extension spec "E-math" (DESIGN.md): + - * / sqrt floor and integer operations only.

RM_HD uint64_t f2u(double d) { return __builtin_bit_cast(uint64_t, d); }
RM_HD double u2f(uint64_t u) { return __builtin_bit_cast(double, u); }

RM_HD double det_log2(double x)
{
    const uint64_t b = f2u(x);
    int e = int((b >> 52) & 0x7FFu) - 1023;
    double m = u2f((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    const double f = (m - 1.0) / (m + 1.0);
    const double s = f * f;
    // atanh series: sum_{k=1..11} s^k / (2k+1), highest order first
    double q = 1.0 / 23.0;
#pragma unroll
    for (int k = 10; k >= 1; --k) {
        q = q * s + 1.0 / double(2 * k + 1);
    }
    q = q * s;
    const double r = f + f * q;
    return double(e) + (2.0 * r) * 1.4426950408889634;
}

RM_HD double det_exp2(double y)
{
    if (y != y) return y;
    if (!(y >= -1022.0)) return 0.0;
    if (y > 1023.0) return u2f(0x7FF0000000000000ull);
    const double k = floor(y + 0.5);
    const double r = y - k;
    const double t = r * 0.6931471805599453;
    // exp(t) = sum t^n / n!, n = 13 .. 0
    const double inv_fact[14] = {1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0,
                                 1.0 / 40320.0, 1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0,
                                 1.0 / 479001600.0, 1.0 / 6227020800.0};
    double q = inv_fact[13];
#pragma unroll
    for (int n = 12; n >= 0; --n) {
        q = q * t + inv_fact[n];
    }
    const double scale = u2f(uint64_t(int64_t(k) + 1023) << 52);
    return q * scale;
}

RM_HD double det_log10(double x) { return det_log2(x) * 0.30102999566398120; }
RM_HD double det_pow10(double y) { return det_exp2(y * 3.3219280948873622); }

// Acklam's rational approximation of the standard normal quantile
RM_HD double det_normal(double u)
{
    const double a1 = -3.969683028665376e+01, a2 = 2.209460984245205e+02, a3 = -2.759285104469687e+02,
                 a4 = 1.383577518672690e+02, a5 = -3.066479806614716e+01, a6 = 2.506628277459239e+00;
    const double b1 = -5.447609879822406e+01, b2 = 1.615858368580409e+02, b3 = -1.556989798598866e+02,
                 b4 = 6.680131188771972e+01, b5 = -1.328068155288572e+01;
    const double c1 = -7.784894002430293e-03, c2 = -3.223964580411365e-01, c3 = -2.400758277161838e+00,
                 c4 = -2.549732539343734e+00, c5 = 4.374664141464968e+00, c6 = 2.938163982698783e+00;
    const double d1 = 7.784695709041462e-03, d2 = 3.224671290700398e-01, d3 = 2.445134137142996e+00,
                 d4 = 3.754408661907416e+00;
    // The two tails are one code path (a wave almost always holds lanes of both, and divergent
    // branches are issued one after the other): the upper tail is the lower tail's expression on
    // 1 - u with the sign flipped -- operation for operation what the extension spec writes as two cases.
    const bool lower = u < 0.02425;
    if (lower || !(u <= 0.97575)) {
        const double t = lower ? u : 1.0 - u;
        const double q = sqrt(-2.0 * (det_log2(t) * 0.6931471805599453));
        const double r = (((((c1 * q + c2) * q + c3) * q + c4) * q + c5) * q + c6) /
                         ((((d1 * q + d2) * q + d3) * q + d4) * q + 1.0);
        return lower ? r : -r;
    }
    const double q = u - 0.5;
    const double r = q * q;
    return (((((a1 * r + a2) * r + a3) * r + a4) * r + a5) * r + a6) * q /
           (((((b1 * r + b2) * r + b3) * r + b4) * r + b5) * r + 1.0);
}

RM_HD uint64_t mix64(uint64_t z)
{
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

double host_det_pow10(double y) { return det_pow10(y); }
uint64_t host_mix64(uint64_t z) { return mix64(z); }

// per-link shadowing deviate: symmetric in (a, b), independent of evaluation order / sharding
RM_HD double shadow_gauss(uint64_t seed_mixed, double clip, uint32_t a, uint32_t b)
{
    const uint32_t lo = a < b ? a : b;
    const uint32_t hi = a < b ? b : a;
    const uint64_t h = mix64(seed_mixed ^ ((uint64_t(lo) << 32) | uint64_t(hi)));
    const double u = (double(h >> 12) + 0.5) * 0x1.0p-52;
    double g = det_normal(u);
    if (g > clip) g = clip;
    if (g < -clip) g = -clip;
    return g;
}

// Position.getDistance, Position.java:56-64: this = transmitter, p2 = receiver;
// (dx*dx + dy*dy) + dz*dz, then a correctly rounded square root.
RM_HD double ref_distance(double ax, double ay, double az, double bx, double by, double bz)
{
    double dx = ax - bx;
    double dy = ay - by;
    double dz = az - bz;
    dx = dx * dx;
    dy = dy * dy;
    dz = dz * dz;
    return sqrt(dx + dy + dz);
}

RM_HD double logdist_rssi(const ModelDev &m, const rm_tx_record &tx, double rx, double ry, double rz, int j)
{
    const double d = ref_distance(tx.x, tx.y, tx.z, rx, ry, rz);
    const double dd = (d > m.ld_d0) ? d : m.ld_d0;
    const double t1 = tx.txpower - m.ld_pl0;
    const double t2 = 10.0 * m.ld_exp;
    const double l = det_log10(dd / m.ld_d0);
    double rssi = t1 - t2 * l;
    if (m.ld_sigma > 0.0) {
        rssi = rssi - m.ld_sigma * shadow_gauss(m.ld_seed_mixed, m.ld_clip, uint32_t(tx.src), uint32_t(j));
    }
    return rssi;
}

// ---- Q80 fixed point (exact, order-independent interference sums) ------------------------
struct U128 {
    uint64_t lo, hi;
};

RM_HD U128 u128_add(U128 a, U128 b)
{
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}

RM_HD U128 q80_from_double(double lin)
{
    U128 r = {0, 0};
    if (!(lin > 0.0)) return r;
    const uint64_t bits = f2u(lin);
    const int ex = int((bits >> 52) & 0x7FFu);
    if (ex == 0x7FF) {
        r.lo = ~0ull;
        r.hi = 0x7FFFFFFFFFFFFFFFull;
        return r;
    }
    if (ex == 0) return r;
    const uint64_t man = (bits & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
    const int shift = ex - 1075 + 80;
    if (shift >= 0) {
        if (shift > 74) {
            r.lo = ~0ull;
            r.hi = 0x7FFFFFFFFFFFFFFFull;
            return r;
        }
        if (shift >= 64) {
            r.hi = man << (shift - 64);
        } else if (shift == 0) {
            r.lo = man;
        } else {
            r.lo = man << shift;
            r.hi = man >> (64 - shift);
        }
        return r;
    }
    if (-shift >= 64) return r;
    r.lo = man >> (-shift);
    return r;
}

RM_HD int clz64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)v);
#else
    return __builtin_clzll(v);
#endif
}

RM_HD double q80_to_double(U128 q)
{
    if (q.hi == 0 && q.lo == 0) return 0.0;
    const int top = q.hi ? 127 - clz64(q.hi) : 63 - clz64(q.lo);
    uint64_t keep;
    int drop = 0;
    if (top <= 52) {
        keep = q.lo;
    } else {
        drop = top - 52;
        // keep = q >> drop ; rem = q & ((1<<drop)-1)
        uint64_t rem_hi, rem_lo, half_hi, half_lo;
        if (drop >= 64) {
            keep = q.hi >> (drop - 64);
            rem_hi = (drop == 64) ? 0 : (q.hi & ((1ull << (drop - 64)) - 1));
            rem_lo = q.lo;
            half_hi = (drop == 64) ? 0 : (1ull << (drop - 65));
            half_lo = (drop == 64) ? (1ull << 63) : 0;
        } else {
            keep = (q.lo >> drop) | (q.hi << (64 - drop));
            rem_hi = 0;
            rem_lo = q.lo & ((1ull << drop) - 1);
            half_hi = 0;
            half_lo = 1ull << (drop - 1);
        }
        const bool gt = (rem_hi > half_hi) || (rem_hi == half_hi && rem_lo > half_lo);
        const bool eq = (rem_hi == half_hi) && (rem_lo == half_lo);
        if (gt || (eq && (keep & 1ull))) keep += 1;
    }
    // keep * 2^(drop-80): both factors exact
    const double scale = u2f(uint64_t(drop - 80 + 1023) << 52);
    return double(keep) * scale;
}

// ---- java.util.Random (Java SE specification) ----------------------------------------------
constexpr uint64_t kLcgA = 0x5DEECE66Dull;
constexpr uint64_t kLcgC = 0xBull;
constexpr uint64_t kLcgMask = (1ull << 48) - 1;

// affine map of `steps` LCG steps: s -> A*s + C (mod 2^48)
RM_HD void lcg_jump_map(uint64_t steps, uint64_t &A, uint64_t &C)
{
    uint64_t a = kLcgA, c = kLcgC;
    uint64_t accA = 1, accC = 0;
    while (steps) {
        if (steps & 1ull) {
            accA = (accA * a) & kLcgMask;
            accC = (accC * a + c) & kLcgMask;
        }
        c = ((a + 1) * c) & kLcgMask;
        a = (a * a) & kLcgMask;
        steps >>= 1;
    }
    A = accA;
    C = accC;
}

void host_lcg_jump_map(uint64_t steps, uint64_t *A, uint64_t *C) { lcg_jump_map(steps, *A, *C); }

RM_HD double lcg_next_double(uint64_t &s)
{
    s = (s * kLcgA + kLcgC) & kLcgMask;
    const int64_t hi = int64_t(s >> 22); // next(26)
    s = (s * kLcgA + kLcgC) & kLcgMask;
    const int64_t lo = int64_t(s >> 21); // next(27)
    return double((hi << 27) + lo) * 0x1.0p-53;
}

// ============================================================================ small kernels

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
    if (lane == 0) {
        nd.bbox_xy[g] = make_float4(lox, loy, hix, hiy);
        nd.bbox_z[g] = make_float2(loz, hiz);
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
    for (int g = b * 16; g < min(n_groups, b * 16 + 16); ++g) {
        const float4 q = nd.bbox_xy[g];
        const float2 qz = nd.bbox_z[g];
        xy.x = fminf(xy.x, q.x);
        xy.y = fminf(xy.y, q.y);
        xy.z = fmaxf(xy.z, q.z);
        xy.w = fmaxf(xy.w, q.w);
        z.x = fminf(z.x, qz.x);
        z.y = fmaxf(z.y, qz.y);
    }
    nd.wg_box_xy[b] = xy;
    nd.wg_box_z[b] = z;
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
RM_D void tx_prefilter(const ModelDev &m, const rm_tx_record &tx, float4 &f, double &thr64)
{
    const double inf = u2f(0x7FF0000000000000ull);
    double cut; // cut-off distance (metres): no link beyond it can matter
    if (tx.src < 0) {
        cut = -1.0; // padding record
    } else if (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_UDGM_CONST) {
        cut = m.geo_cut;
    } else if (m.kind == RM_MODEL_LOGDIST) {
        const double margin = tx.txpower - m.ld_pl0 + m.ld_sigma * m.ld_clip - (m.ld_level - 1e-6);
        if (!(margin >= 0.0)) {
            cut = -1.0;
        } else if (!(m.ld_exp > 0.0)) {
            cut = inf;
        } else {
            // hardware fp32 exp2 (relative error ~1e-6 at these arguments) with a 1e-4 pad
            cut = m.ld_d0 * double(__builtin_amdgcn_exp2f(float(margin / (10.0 * m.ld_exp) * 3.3219280948873622))) * (1.0 + 1e-4);
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
        r.x = nd.sx[s];
        r.y = nd.sy[s];
        r.z = nd.sz[s];
        r.txpower = nd.stxpower[s];
        r.txprob = nd.stxprob[s];
        r.start_us = start_us;
        r.air_us = air_us;
        r.src = s;
        r.channel = nd.schannel[s];
    }
    return r;
}

__global__ void __launch_bounds__(256)
k_pack_tx(NodesDev nd, const int32_t *src, int n, int64_t start_us, int64_t air_us, rm_tx_record *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rm_tx_record r = make_tx_record(nd, src[i], start_us, air_us);
    out[i] = r;
}

struct PackStarts {
    int64_t start_us[kMaxBatch];
};

__global__ void __launch_bounds__(256)
k_pack_tx_batch(NodesDev nd, const int32_t *src, int n, PackStarts st, int64_t air_us, rm_tx_record *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t o = size_t(blockIdx.y) * n + i;
    const rm_tx_record r = make_tx_record(nd, src[o], st.start_us[blockIdx.y], air_us);
    out[o] = r;
}

// ============================================================================ the filter kernel

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
    if (slab >= t.n_slabs) return;

    // counters that later kernels of this tick (cursor) or the next tick's filter (candidate
    // totals, other parity) add to start at zero
    if (t.use_matrix) {
        if (e0 >= t.cnt_base) t.cnt[(size_t((e0 - t.cnt_base) / kTxChunk) * t.n_slabs + slab) * 64 + lane] = 0u;
    } else if (blockIdx.x == 0) {
        for (int i = blockIdx.y * kBlock + threadIdx.x; i < t.zero_len; i += gridDim.y * kBlock) {
            t.cursor[i] = 0u;
            t.cand_tot_next[i] = 0u;
        }
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
                t.st_blk[idx] = fbase;
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
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_eval) return;
    const int abs_i = t.first_eval + e;
    rm_tx_record tx;
    if (t.src_list && abs_i >= t.first_new) {
        tx = make_tx_record(nd, t.src_list[abs_i - t.first_new], t.src_start_us, t.src_air_us);
        t.tx_build[abs_i] = tx;
    } else {
        tx = t.tx[abs_i];
    }
    float4 f;
    double thr64;
    tx_prefilter(m, tx, f, thr64);
    float inv = 0.f;
    if (m.shadow_tbl && f.w > 0.f && f.w < __builtin_inff()) {
        const double cut = sqrt(double(f.w));
        if (2.0 * m.f32_slack / (0.15 * cut) + 1e-5 <= kShadowPad) inv = float(kShadowBins) / f.w;
    }
    t.p_txf[e] = f;
    t.p_ch[e] = tx.channel;
    t.p_src[e] = tx.src;
    t.p_inv[e] = inv;
}

__global__ void __launch_bounds__(256) k_tick_prep(const NodesDev nd, const ModelDev m, const TickDev t)
{
    tick_prep_body(nd, m, t);
}

template <int RPT, bool SHADOW>
RM_D void filter_wg_body(const NodesDev &nd, const ModelDev &m, const TickDev &t)
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
    const int n_groups = (t.n_rx + kGroup - 1) / kGroup;

    if (SHADOW) s_tbl[threadIdx.x] = m.shadow_tbl[threadIdx.x];
    if (threadIdx.x == 0) s_n = 0u;

    // phase A works on kUnrollA x 256 frames at a time: their records are requested together (one
    // exposed round trip per 1024 frames), the first ones before anything else
    constexpr int kUnrollA = 4;
    float4 af[kUnrollA];
    int ach[kUnrollA], asrc[kUnrollA];
    float ainv[kUnrollA];
    auto request = [&](int g0) {
#pragma unroll
        for (int u = 0; u < kUnrollA; ++u) {
            const int e = g0 + u * kBlock + int(threadIdx.x);
            af[u] = make_float4(0.f, 0.f, 0.f, -1.f);
            ach[u] = 0;
            asrc[u] = -1;
            ainv[u] = 0.f;
            if (e < n_eval) {
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

    // this wave's receivers, resident in registers for the whole tick
    float fx[RPT], fy[RPT], fz[RPT];
    int fch[RPT], forig[RPT];
    float4 bxy[RPT];
    float2 bz[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int j = jbase + r * kGroup + lane;
        fx[r] = fy[r] = fz[r] = __builtin_nanf("");
        fch[r] = 0;
        forig[r] = 0;
        if (live && j < t.n_rx) {
            const float4 v = nd.rxf[j];
            fx[r] = v.x;
            fy[r] = v.y;
            fz[r] = v.z;
            fch[r] = __float_as_int(v.w);
            if (SHADOW) forig[r] = nd.orig[j];
        }
        const int g = slab * RPT + r;
        const bool ok = live && g * kGroup < t.n_rx;
        bxy[r] = ok ? nd.bbox_xy[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        bz[r] = ok ? nd.bbox_z[g] : make_float2(0.f, 0.f);
    }

    // union box of the workgroup's 4*RPT groups
    float4 wxy;
    float2 wz;
    if (RPT == 4) {
        wxy = nd.wg_box_xy[wg];
        wz = nd.wg_box_z[wg];
    } else {
        const float inf_ = __builtin_inff();
        wxy = make_float4(inf_, inf_, -inf_, -inf_);
        wz = make_float2(inf_, -inf_);
        for (int g = wg * kWavesPerBlock * RPT; g < min(n_groups, (wg + 1) * kWavesPerBlock * RPT); ++g) { // uniform
            const float4 q = nd.bbox_xy[g];
            const float2 qz = nd.bbox_z[g];
            wxy.x = fminf(wxy.x, q.x);
            wxy.y = fminf(wxy.y, q.y);
            wxy.z = fmaxf(wxy.z, q.z);
            wxy.w = fmaxf(wxy.w, q.w);
            wz.x = fminf(wz.x, qz.x);
            wz.y = fmaxf(wz.y, qz.y);
        }
    }
    __syncthreads();

    uint32_t round = 0;
    for (int g0 = 0; g0 < n_eval; g0 += kUnrollA * kBlock) { // block-uniform
    if (g0) request(g0);
#pragma unroll 1
    for (int u = 0; u < kUnrollA; ++u) { // rolled: one copy of phase B; the records are selected, not indexed
        const int f0 = g0 + u * kBlock;
        if (f0 >= n_eval) break; // block-uniform
        // phase A: this thread's frame against the workgroup box
        const int e = f0 + int(threadIdx.x);
        float4 tfa = af[0];
        int cha = ach[0], srca = asrc[0];
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
        }
        bool hit = false;
        if (e < n_eval) {
            const float dx = fmaxf(fmaxf(wxy.x - tfa.x, tfa.x - wxy.z), 0.f);
            const float dy = fmaxf(fmaxf(wxy.y - tfa.y, tfa.y - wxy.w), 0.f);
            const float dz = fmaxf(fmaxf(wz.x - tfa.z, tfa.z - wz.y), 0.f);
            hit = dist2_f32(dx, dy, dz) <= tfa.w;
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
        const bool last = f0 + kBlock >= n_eval;
        if (!last && n_near + kBlock <= kNearLds) continue; // room for another 256 frames

        // phase B: chunks of 64 near frames, every wave for its own groups
        if (live) {
            for (int c0 = 0; c0 < n_near; c0 += kTxChunk) {
                const int nt = min(kTxChunk, n_near - c0);
                uint64_t near[RPT];
                uint64_t todo = 0;
                {
                    const float4 tf = s_txf[c0 + min(lane, nt - 1)];
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        near[r] = 0;
                        if ((slab * RPT + r) * kGroup < t.n_rx) {
                            const float dx = fmaxf(fmaxf(bxy[r].x - tf.x, tf.x - bxy[r].z), 0.f);
                            const float dy = fmaxf(fmaxf(bxy[r].y - tf.y, tf.y - bxy[r].w), 0.f);
                            const float dz = fmaxf(fmaxf(bz[r].x - tf.z, tf.z - bz[r].y), 0.f);
                            near[r] = ballot64(lane < nt && dist2_f32(dx, dy, dz) <= tf.w);
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
                const uint32_t shard = (uint32_t(slab) + (round + uint32_t(c0 >> 6)) * 37u + blockIdx.z * 101u) & t.shard_mask;
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
                            t.st_blk[idx] = fbase;
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
}

template <int RPT, bool SHADOW>
__global__ void __launch_bounds__(kBlock, RPT == 4 ? 4 : (RPT == 2 ? 5 : 6)) k_filter_wg(const NodesDev nd, const ModelDev m, const TickDev t)
{
    filter_wg_body<RPT, SHADOW>(nd, m, t);
}

// ============================================================================ exact evaluation

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
constexpr int kFusedScanMax = 8192;
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

// One lane per candidate link (full waves): the reference's fp64 arithmetic.
// SEG 0: unsorted table, heard links are counted per (frame, slab) cell (ordered scatter later).
// SEG 1/2: sorted table, heard links go straight into the frame's segment of the A records (any
// order inside it); the segment offsets are the scan of the per-frame candidate counts, redone in
// LDS by every workgroup (1; 3 / 4 = the same for at most kSmallScan / kMediumScan frames) or read from k_scan_counts' output (2).
// PACKED: a 1-D grid whose workgroups walk the started 256-entry chunks of all shards (no workgroup
// without entries); otherwise blockIdx.y is the shard and blockIdx.x strides over its entries.
template <int MODEL, bool SINR, bool STOCH, int SEG, bool PACKED = false>
RM_D void exact_body(const NodesDev &nd, const ModelDev &m, const TickDev &t)
{
    __shared__ uint32_t s_seg[scan_lds(SEG)];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_cs[PACKED ? kShards + 1 : 1]; // exclusive scan of the shards' chunk counts
    __shared__ uint32_t s_sn[PACKED ? kShards : 1];     // entries per shard
    const bool publisher = (blockIdx.x == 0 && blockIdx.y == 0);
    bool scanned = false; // the scan is only needed by the scatter: it runs after the first evaluation
    const uint32_t n_own = min(t.shard_count[(PACKED ? threadIdx.x : blockIdx.y) * kShardStride], t.seg_cap); // kBlock == kShards
    constexpr bool kRegScan = (SEG == 3 || SEG == 4);
    SmallCounts<scan_per(SEG)> pre{};
    if (kRegScan && !PACKED && (blockIdx.x == 0 || blockIdx.x * blockDim.x < n_own)) pre = small_scan_load<scan_per(SEG)>(t.cand_tot, t.n_cnt);
    const uint32_t stride = gridDim.x * blockDim.x;
    const int per_slab = kGroup * t.rpt;
    const int lane = threadIdx.x & 63;
    uint32_t n_chunks = 0;
    if (PACKED) {
        const uint32_t mine = (n_own + 255u) >> 8;
        uint32_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) s_wave[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t run = inc - mine;
        for (int w = 0; w < int(threadIdx.x >> 6); ++w) run += s_wave[w];
        s_cs[threadIdx.x] = run;
        s_sn[threadIdx.x] = n_own;
        n_chunks = uniform_u(s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3]);
        if (threadIdx.x == 0) s_cs[PACKED ? kShards : 0] = n_chunks;
        __syncthreads();
        // a workgroup without a chunk leaves before touching the per-frame counts (a receiver
        // partition has few candidates per tick: most of the grid); the publisher stays for seg_off
        if (blockIdx.x >= n_chunks && !publisher) return;
        if (kRegScan) pre = small_scan_load<scan_per(SEG)>(t.cand_tot, t.n_cnt);
    }
    auto entries = [&](const uint32_t shard, const uint32_t it, const uint32_t n) {
        const uint32_t i = it + threadIdx.x;
        const bool valid = i < n;
        const uint32_t idx = shard * t.seg_cap + i;
        bool wanted = false;
        int slot = -1, key = -1;
        uint8_t fl = 0;
        double rssi = 0.0, prob = 1.0;
        int orig = 0;
        if (valid) {
            const int erel = t.st_pkt[idx];
            const int pos = t.st_dst[idx];
            const rm_tx_record tx = t.tx[t.first_eval + erel];
            const bool is_new = (t.first_eval + erel) >= t.first_new;
            const RxRecord rx_ = nd.rec[pos];
            const LinkEval ev = eval_link<MODEL, SINR>(m, nd, tx, rx_, is_new);
            fl = ev.append ? ev.flags : uint8_t(0);
            if (ev.append && MODEL != RM_MODEL_NULL && MODEL != RM_MODEL_UDGM_CONST && tx_success(m, tx) <= 0.0) fl |= kFlagTxDead;
            t.st_flags[idx] = fl;
            if (ev.append) {
                orig = rx_.orig;
                if (MODEL == RM_MODEL_LOGDIST) {
                    rssi = ev.aux;
                    prob = rx_.rxprob;
                } else {
                    rssi = tx.txpower; // reference media hand the packet's transmit power through
                    prob = (MODEL == RM_MODEL_UDGM || MODEL == RM_MODEL_N2N) ? ev.aux : 1.0;
                }
                if (SEG == 0 || SINR) { // the ordered scatter / the SINR pass read these from the entry
                    t.st_orig[idx] = orig;
                    t.st_aux[idx] = rssi;
                    t.st_prob[idx] = prob;
                }
                if (SINR) {
                    t.st_lin[idx] = ev.lin;
                    t.st_next[idx] = atomicExch(&t.head[pos], int(idx));
                }
                wanted = ev.wanted;
            }
            slot = erel - t.cnt_base;
            key = (SEG == 0) ? int((size_t(slot >> 6) * t.n_slabs + pos / per_slab) * 64 + (slot & 63)) : slot;
        }
        if (SEG == 1 && !scanned) { // block-uniform
            block_scan_counts(t.cand_tot, t.n_cnt, s_seg, s_wave, publisher ? t.seg_off : nullptr, nullptr);
            scanned = true;
        }
        if (kRegScan && !scanned) {
            small_scan(pre, t.n_cnt, s_seg, s_wave, publisher ? t.seg_off : nullptr, nullptr);
            scanned = true;
        }
        // one atomic per run of same-frame (same-cell) entries
        const RunInfo ri = run_prefix(key, wanted, lane);
        if (SEG == 0) {
            if (valid && lane == ri.start && ri.total) atomicAdd(&t.cnt[key], ri.total);
        } else {
            uint32_t base = 0;
            if (valid && lane == ri.start && ri.total) base = atomicAdd(&t.cursor[slot], ri.total);
            base = __shfl(base, ri.start);
            if (wanted) {
                const uint32_t o = ((SEG == 1 || kRegScan) ? s_seg[slot] : t.seg_off[slot]) + base + ri.before;
                t.a_dst[o] = orig;
                t.a_rssi[o] = rssi;
                if (SINR) t.a_e[o] = int(idx);
                if (STOCH) {
                    t.a_prob[o] = prob;
                    t.a_verdict[o] = uint8_t(0); // pending: k_apply_draws decides
                } else {
                    t.a_verdict[o] = (fl & kFlagTxDead) ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
                }
            }
        }
    };
    if (PACKED) {
        for (uint32_t u = blockIdx.x; u < n_chunks; u += gridDim.x) { // block-uniform
            uint32_t lo = 0, hi = kShards; // the shard whose chunk range holds u: s_cs[lo] <= u < s_cs[lo + 1]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (uniform_u(s_cs[mid]) <= u) lo = mid; else hi = mid;
            }
            entries(lo, (u - uniform_u(s_cs[lo])) << 8, uniform_u(s_sn[lo]));
        }
    } else {
        for (uint32_t it = blockIdx.x * blockDim.x; it < n_own; it += stride) entries(blockIdx.y, it, n_own); // block-uniform trip count
    }
    if (SEG == 1 && !scanned && publisher) // seg_off is published even if this shard was empty
        block_scan_counts(t.cand_tot, t.n_cnt, s_seg, s_wave, t.seg_off, nullptr);
    if (kRegScan && !scanned && publisher) small_scan(pre, t.n_cnt, s_seg, s_wave, t.seg_off, nullptr);
}

template <int MODEL, bool SINR, bool STOCH, int SEG>
__global__ void __launch_bounds__(256) k_exact(const NodesDev nd, const ModelDev m, const TickDev t)
{
    exact_body<MODEL, SINR, STOCH, SEG>(nd, m, t);
}

// Several independent ticks per launch (rm_batch_*): blockIdx.z selects the tick.  The ticks'
// descriptors sit in device memory (read with scalar loads at a uniform address); k_store_ticks
// writes them there from its kernel arguments, kStoreTicks at a time (4 KB of arguments).
constexpr int kStoreTicks = 6;
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

template <int MODEL, bool STOCH, int SCAN>
__global__ void __launch_bounds__(256) k_exact_batch(const NodesDev nd, const ModelDev m, const TickDev *__restrict__ ticks)
{
    exact_body<MODEL, false, STOCH, SCAN, true>(nd, m, ticks[blockIdx.z]);
}

// half duplex (SINR mode): every frame on the air leaves a SELF entry in its source's list
__global__ void __launch_bounds__(256) k_self_entries(NodesDev nd, TickDev t)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_eval = t.n_active - t.first_eval;
    if (e >= n_eval) return;
    const int src = t.tx[t.first_eval + e].src;
    if (src < nd.rx_first || src >= nd.rx_first + nd.n_rx) return;
    const int pos = nd.pos_of[src - nd.rx_first];
    const uint32_t shard = (blockIdx.x * 4 + (threadIdx.x >> 6)) & t.shard_mask;
    const uint32_t local = atomicAdd(&t.shard_count[shard * kShardStride], 1u);
    if (local >= t.seg_cap) {
        t.stage_count[1] = 1u;
        return;
    }
    const uint32_t idx = shard * t.seg_cap + local;
    t.st_pkt[idx] = e;
    t.st_dst[idx] = pos;
    t.st_blk[idx] = idx;
    t.st_flags[idx] = kFlagSelf;
    t.st_aux[idx] = 0.0;
    t.st_prob[idx] = 0.0;
    t.st_orig[idx] = src;
    t.st_lin[idx] = 0.0;
    t.st_next[idx] = atomicExch(&t.head[pos], int(idx));
}

// ============================================================================ offsets (tiny scans)

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

// off[cell] = heard links of the same frame in lower slabs; slot_tot[slot] = heard links of the
// frame.  One 1024-thread workgroup per tile of 64 frames: lane = frame, each wave owns a
// contiguous range of slabs (coalesced 256-byte rows).
__global__ void __launch_bounds__(1024) k_cell_off(TickDev t)
{
    __shared__ uint32_t s_part[16][64];
    const int cc = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (t.n_slabs + 15) / 16;
    const int s0 = min(t.n_slabs, wave * per), s1 = min(t.n_slabs, s0 + per);
    const size_t row0 = size_t(cc) * t.n_slabs;
    uint32_t sum = 0;
    for (int s = s0; s < s1; ++s) sum += t.cnt[(row0 + s) * 64 + lane];
    s_part[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t x = s_part[w][lane];
        if (w < wave) run += x;
        total += x;
    }
    for (int s = s0; s < s1; ++s) {
        const size_t c = (row0 + s) * 64 + lane;
        t.off[c] = run;
        run += t.cnt[c];
    }
    if (wave == 0) t.slot_tot[cc * 64 + lane] = total;
}

// slot_off = exclusive scan over frames of their heard-link totals; publishes the link count
__global__ void __launch_bounds__(1024) k_slot_scan(TickDev t)
{
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    uint32_t vmax = 0;
    for (int base = 0; base < t.n_cnt; base += 1024) {
        const int slot = base + threadIdx.x;
        const uint32_t v = (slot < t.n_cnt) ? t.slot_tot[slot] : 0u;
        vmax = max(vmax, v);
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
        if (slot < t.n_cnt) t.slot_off[slot] = carry + ex;
        carry += total;
    }
    for (int d = 32; d >= 1; d >>= 1) vmax = max(vmax, uint32_t(__shfl_xor(int(vmax), d)));
    if ((threadIdx.x & 63) == 0 && vmax) atomicMax(&t.out_count[3], vmax);
    if (threadIdx.x == 0) {
        t.slot_off[t.n_cnt] = carry;
        t.out_count[0] = carry < t.cap ? carry : t.cap;
        t.out_count[1] = (carry > t.cap || t.stage_count[1] != 0u) ? 1u : 0u;
        t.out_count[2] = carry;
    }
}

// ============================================================================ SINR (O(heard links))

RM_D bool frames_overlap(const rm_tx_record &w, const rm_tx_record &k)
{
    return k.start_us < w.start_us + w.air_us && k.start_us + k.air_us > w.start_us;
}

// one thread per link entry that is a heard link of a new frame: walk the receiver's list,
// sum the co-channel, time-overlapping interferers exactly (Q80), apply capture + half duplex
__global__ void __launch_bounds__(256) k_sinr(ModelDev m, TickDev t)
{
    const uint32_t n = min(t.shard_count[blockIdx.y * kShardStride], t.seg_cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t e = blockIdx.y * t.seg_cap + i;
        if (!(t.st_flags[e] & kFlagHeardNew)) continue;
        const int pos = t.st_dst[e];
        const rm_tx_record w = t.tx[t.first_eval + t.st_pkt[e]];
        U128 acc = {0, 0};
        bool half_duplex = false;
        for (int idx = t.head[pos]; idx >= 0; idx = t.st_next[idx]) {
            if (uint32_t(idx) == e) continue;
            const rm_tx_record k = t.tx[t.first_eval + t.st_pkt[idx]];
            if (!frames_overlap(w, k)) continue;
            const uint8_t fl = t.st_flags[idx];
            if (fl & kFlagSelf) {
                half_duplex = true;
                continue;
            }
            if (!(fl & kFlagInterferer)) continue;
            acc = u128_add(acc, q80_from_double(t.st_lin[idx]));
        }
        const double I = q80_to_double(acc);
        const double denom = I + m.ld_noise_lin;
        const double sinr = t.st_aux[e] - 10.0 * det_log10(denom);
        t.st_sinr[e] = sinr;
        t.st_coll[e] = (half_duplex || !(sinr >= m.ld_capture)) ? 1 : 0;
    }
}

// ============================================================================ ordered scatter

// frames beyond what the fused scans hold: seg_off / slot_off from a one-workgroup scan kernel
__global__ void __launch_bounds__(1024) k_scan_counts(const uint32_t *cnt, uint32_t *off, int n)
{
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < n) ? cnt[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
        if (i < n) off[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) off[n] = carry;
}

// Unsorted tables (Null / N2N media): link entry -> cell offset + rank inside the (frame, slab)
// block -- directly the final (packet, node index) order; verdict for everything that needs no draw.
template <bool STOCH>
__global__ void __launch_bounds__(256) k_finalize(ModelDev m, TickDev t)
{
    const uint32_t n = min(t.shard_count[blockIdx.y * kShardStride], t.seg_cap);
    const uint32_t stride = gridDim.x * blockDim.x;
    const int per_slab = kGroup * t.rpt;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t e = blockIdx.y * t.seg_cap + i;
        const uint8_t fl = t.st_flags[e];
        if (!(fl & kFlagHeardNew)) continue;
        const int slot = t.st_pkt[e] - t.cnt_base;
        const int slab = t.st_dst[e] / per_slab;
        uint32_t rank = 0;
        for (uint32_t k = t.st_blk[e]; k < e; ++k) rank += (t.st_flags[k] & kFlagHeardNew) ? 1u : 0u;
        const uint32_t o = t.slot_off[slot] + t.off[(size_t(slot >> 6) * t.n_slabs + slab) * 64 + (slot & 63)] + rank;
        if (o >= t.cap) continue;
        t.out_pkt[o] = slot - t.shift;
        t.out_dst[o] = t.st_orig[e];
        t.out_rssi[o] = t.st_aux[e];
        const bool sinr = (m.kind == RM_MODEL_LOGDIST) && (m.flags & RM_LD_SINR);
        t.out_sinr[o] = sinr ? t.st_sinr[e] : 0.0;
        const bool collided = sinr && t.st_coll[e];
        if (STOCH) {
            t.out_prob[o] = t.st_prob[e];
            t.out_verdict[o] = collided ? uint8_t(RM_INTERFERED) : uint8_t(0); // 0 = pending
        } else {
            t.out_verdict[o] = ((fl & kFlagTxDead) || collided) ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
        }
    }
}

// per-packet Tx-failure flag where no draw can happen (txSuccess <= 0 is the only way to fail)
RM_D void write_pkt_interference(const ModelDev &m, const TickDev &t, uint32_t first, uint32_t stride)
{
    const bool draws_possible = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_N2N || m.kind == RM_MODEL_LOGDIST);
    const int n_new = t.n_active - t.first_new;
    for (uint32_t q = first; q < uint32_t(n_new); q += stride) {
        const rm_tx_record tx = t.tx[t.first_new + q];
        t.pkt_interference[q] = (draws_possible && tx_success(m, tx) <= 0.0) ? 1 : 0;
    }
}

// Sorted tables: the heard links of a frame sit unordered in the frame's segment of the A
// records.  One wave per frame ranks them by node index -- the order the reference's loop visits
// receivers in (UDGMRadioMedium.java:99) -- and writes them to their final, compact place.
// A segment of up to 64 links sits one per lane and is ranked with a readlane loop, longer ones by
// counting through memory.  MODE 1: the scan of the per-frame heard counts is redone in every
// workgroup (LDS); MODE 2: slot_off comes from k_scan_counts.
template <bool STOCH, bool SINR, int MODE>
RM_D void reorder_body(const ModelDev &m, const TickDev &t)
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
    const int q0 = blockIdx.x * 4 + wave_index();
    uint32_t src0 = 0, len = 0;
    int mine = 0x7fffffff, in_e = 0;
    double in_rssi = 0.0, in_prob = 1.0;
    uint8_t v = 0;
    if (q0 < n_new) {
        src0 = uniform_u(t.seg_off[q0 + t.shift]);
        len = uniform_u(t.cursor[q0 + t.shift]);
        if (uint32_t(lane) < len) {
            const uint32_t o = src0 + lane;
            mine = t.a_dst[o];
            in_rssi = t.a_rssi[o];
            v = t.a_verdict[o];
            if (STOCH) in_prob = t.a_prob[o];
            if (SINR) in_e = t.a_e[o];
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
    } else if (publisher && threadIdx.x == 0) {
        const uint32_t total = t.slot_off[t.n_cnt];
        t.out_count[0] = total < t.cap ? total : t.cap;
        t.out_count[1] = (total > t.cap || t.stage_count[1] != 0u) ? 1u : 0u;
        t.out_count[2] = total;
    }

    for (int q = q0; q < n_new; q += gridDim.x * 4) { // wave-uniform
        const int slot = q + t.shift;
        if (q != q0) {
            src0 = uniform_u(t.seg_off[slot]);
            len = uniform_u(t.cursor[slot]);
        }
        const uint32_t dst0 = uniform_u((MODE == 1 || kRegScan) ? s_off[slot] : t.slot_off[slot]);
        for (uint32_t c0 = 0; c0 < len; c0 += 64) {
            const uint32_t o = src0 + c0 + lane;
            const bool valid = c0 + lane < len;
            if (q != q0 || c0 != 0) { // everything but the prefetched first chunk
                mine = valid ? t.a_dst[o] : 0x7fffffff;
                in_rssi = valid ? t.a_rssi[o] : 0.0;
                v = valid ? t.a_verdict[o] : uint8_t(0);
                in_prob = (STOCH && valid) ? t.a_prob[o] : 1.0;
                in_e = (SINR && valid) ? t.a_e[o] : 0;
            }
            uint32_t rank = 0;
            if (len <= 64) {
                for (uint32_t i = 0; i < len; ++i) rank += (__builtin_amdgcn_readlane(mine, int(i)) < mine) ? 1u : 0u;
            } else if (valid) {
                for (uint32_t k = 0; k < len; ++k) rank += (t.a_dst[src0 + k] < mine) ? 1u : 0u;
            }
            const uint32_t d = dst0 + rank;
            if (valid && d < t.cap) {
                t.out_pkt[d] = q;
                t.out_dst[d] = mine;
                t.out_rssi[d] = in_rssi;
                uint8_t vv = v;
                if (SINR) {
                    t.out_sinr[d] = t.st_sinr[in_e];
                    if (t.st_coll[in_e]) vv = RM_INTERFERED;
                } else {
                    t.out_sinr[d] = 0.0;
                }
                t.out_verdict[d] = vv;
                if (STOCH) t.out_prob[d] = in_prob;
            }
        }
    }
    if (!STOCH) write_pkt_interference(m, t, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

template <bool STOCH, bool SINR, int MODE>
__global__ void __launch_bounds__(256) k_reorder(ModelDev m, TickDev t)
{
    reorder_body<STOCH, SINR, MODE>(m, t);
}

template <bool STOCH, int SCAN>
__global__ void __launch_bounds__(256) k_reorder_batch(const ModelDev m, const TickDev *__restrict__ ticks)
{
    reorder_body<STOCH, false, SCAN>(m, ticks[blockIdx.z]);
}

__global__ void __launch_bounds__(256) k_tick_prep_batch(const NodesDev nd, const ModelDev m, const TickDev *__restrict__ ticks)
{
    tick_prep_body(nd, m, ticks[blockIdx.z]);
}

template <int RPT, bool SHADOW>
__global__ void __launch_bounds__(kBlock, RPT == 4 ? 4 : (RPT == 2 ? 5 : 6))
k_filter_wg_batch(const NodesDev nd, const ModelDev m, const TickDev *__restrict__ ticks)
{
    filter_wg_body<RPT, SHADOW>(nd, m, ticks[blockIdx.z]);
}

// ============================================================================ Java-RNG draws

constexpr int kScanTile = 2048; // 256 threads x 8

RM_D uint32_t draw_flag(const TickDev &t, uint32_t pos)
{
    return (t.out_verdict[pos] == 0 && t.out_prob[pos] < 1.0) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) k_draw_tile_sums(TickDev t)
{
    __shared__ uint32_t s_part[4];
    const uint32_t n = t.out_count[0];
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    uint32_t v = 0;
    for (int i = 0; i < 8; ++i) {
        const uint32_t pos = base + i * 256 + threadIdx.x;
        if (pos < n) v += draw_flag(t, pos);
    }
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) t.scan_block[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ void __launch_bounds__(1024) k_draw_tile_scan(TickDev t)
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

__global__ void __launch_bounds__(256) k_draw_scan(TickDev t)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t n = t.out_count[0];
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
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
    uint32_t off = t.scan_block[blockIdx.x] + inc - sum;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (p0 + i < n) t.draw_scan[p0 + i] = off;
        off += f[i];
    }
}

// per-packet number of receiver draws this rank would consume (if the packet's Tx does not fail)
__global__ void __launch_bounds__(256) k_pkt_draw_counts(TickDev t)
{
    const int n_new = t.n_active - t.first_new;
    const uint32_t n = t.out_count[0];
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_new) return;
    const uint32_t b = min(t.slot_off[q + t.shift], n);
    const uint32_t e = min(t.slot_off[q + t.shift + 1], n);
    t.pkt_draw_cnt[q] = t.draw_scan[e] - t.draw_scan[b];
}

// The only sequential part: the shared generator is consumed packet after packet
// (Simulator.getRandom(); UDGMRadioMedium.java:85-92,106).  One workgroup: the per-packet jump
// maps are built in parallel, then one lane walks the packets.  Receiver-sharded ranks all run
// the same chain on the all-gathered per-(rank, packet) draw counts `all_cnt[world][n_new]`: a
// packet's receivers are visited in node order = rank order, so this rank's first draw of packet q
// comes after the draws of the lower ranks, and the generator moves on by the sum over all ranks.
__global__ void __launch_bounds__(1024) k_rng_chain(ModelDev m, TickDev t, const uint32_t *all_cnt, int world, int rank)
{
    __shared__ uint64_t s_A[1024], s_C[1024], s_Ab[1024], s_Cb[1024];
    __shared__ double s_txs[1024];
    __shared__ uint64_t s_state;
    const int n_new = t.n_active - t.first_new;
    if (threadIdx.x == 0) s_state = *t.rng_state & kLcgMask;
    __syncthreads();
    for (int base = 0; base < n_new; base += 1024) {
        const int q = base + threadIdx.x;
        if (q < n_new) {
            uint64_t total = 0, before = 0;
            if (all_cnt) {
                for (int r = 0; r < world; ++r) {
                    const uint32_t v = all_cnt[size_t(r) * n_new + q];
                    if (r < rank) before += v;
                    total += v;
                }
            } else {
                total = t.pkt_draw_cnt[q];
            }
            uint64_t A, C;
            lcg_jump_map(2ull * total, A, C);
            s_A[threadIdx.x] = A;
            s_C[threadIdx.x] = C;
            lcg_jump_map(2ull * before, A, C);
            s_Ab[threadIdx.x] = A;
            s_Cb[threadIdx.x] = C;
            s_txs[threadIdx.x] = tx_success(m, t.tx[t.first_new + q]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t s = s_state;
            const int cnt = min(1024, n_new - base);
            for (int i = 0; i < cnt; ++i) {
                const double txs = s_txs[i];
                bool interference = false;
                if (txs <= 0.0) {
                    interference = true;
                } else if (txs < 1.0) {
                    if (lcg_next_double(s) > txs) interference = true;
                }
                t.pkt_interference[base + i] = interference ? 1 : 0;
                t.pkt_rng[base + i] = (s_Ab[i] * s + s_Cb[i]) & kLcgMask; // this rank's first receiver draw
                if (!interference) s = (s_A[i] * s + s_C[i]) & kLcgMask;
            }
            s_state = s;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *t.rng_state = s_state;
}

__global__ void __launch_bounds__(256) k_apply_draws(TickDev t)
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
        out->sinr[i] = t.out_sinr[i];
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

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

hipError_t launch_prep_rx(hipStream_t s, const NodesDev &nd, const ModelDev &m)
{
    if (nd.n_rx <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_prep_rx, dim3(cdiv(nd.n_rx, kGroup)), dim3(64), 0, s, nd, m);
    const int n_wg = cdiv(nd.n_rx, kGroup * 16);
    hipLaunchKernelGGL(k_wg_boxes, dim3(cdiv(n_wg, 256)), dim3(256), 0, s, nd, n_wg);
    return hipGetLastError();
}

hipError_t launch_pack_tx(hipStream_t s, const NodesDev &nd, const int32_t *dev_src, int n, int64_t start_us,
                          int64_t air_us, rm_tx_record *out)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_tx, dim3(cdiv(n, 256)), dim3(256), 0, s, nd, dev_src, n, start_us, air_us, out);
    return hipGetLastError();
}

// Chooses the filter variant for this tick and fixes the receiver tiling (t.rpt, t.n_slabs):
//  kFilterGrid: k_filter on a (slab, tile) grid -- the general variant (fp64 frame, unsorted tables);
//  kFilterWg:   k_tick_prep + k_filter_wg, two-level cull inside one workgroup per 4*rpt groups.
int plan_filter(TickDev &t, const LaunchCfg &cfg, bool want_wg)
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
        if (const char *e = getenv("RM_FILTER")) {
            if (!strcmp(e, "grid")) mode = kFilterGrid;
            else if (!strcmp(e, "wg")) mode = kFilterWg;
        }
        if (want_wg) mode = kFilterWg;
        if (mode == kFilterWg) {
            // batches bring their own parallelism (workgroups x ticks): the coarse tiling halves the frame x
            // workgroup-box tests of phase A twice over; a lone tick needs the workgroups
            int rpt = (t.n_rx > 400000 || (want_wg && t.n_rx >= 16384)) ? 4 : 1;
            if (const char *e = getenv("RM_WG_RPT")) rpt = (atoi(e) == 4) ? 4 : (atoi(e) == 2 ? 2 : 1);
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
        hipLaunchKernelGGL(k_tick_prep, dim3(cdiv(n_eval, 256)), dim3(256), 0, s, nd, m, t);
        const dim3 grid(cdiv(t.n_slabs, kWavesPerBlock)), block(kBlock);
        if (t.rpt == 4) {
            if (cfg.shadow) hipLaunchKernelGGL((k_filter_wg<4, true>), grid, block, 0, s, nd, m, t);
            else hipLaunchKernelGGL((k_filter_wg<4, false>), grid, block, 0, s, nd, m, t);
        } else if (t.rpt == 2) {
            if (cfg.shadow) hipLaunchKernelGGL((k_filter_wg<2, true>), grid, block, 0, s, nd, m, t);
            else hipLaunchKernelGGL((k_filter_wg<2, false>), grid, block, 0, s, nd, m, t);
        } else {
            if (cfg.shadow) hipLaunchKernelGGL((k_filter_wg<1, true>), grid, block, 0, s, nd, m, t);
            else hipLaunchKernelGGL((k_filter_wg<1, false>), grid, block, 0, s, nd, m, t);
        }
        return hipGetLastError();
    }
    // XCD-aware launch: workgroups are dealt round-robin over the 8 XCDs in linear order (x fastest),
    // so with gridDim.x a multiple of 8 every tile-workgroup of one receiver slab has the same
    // blockIdx.x % 8 -- one XCD, one L2 -- and the slab's records leave HBM once per tick, not once
    // per XCD (placement is a speed matter only; the padding workgroups exit at once)
    const dim3 grid((cdiv(t.n_slabs, kWavesPerBlock) + 7) / 8 * 8, cdiv(n_eval, kTxChunk));
    const dim3 block(kBlock);
#define RM_LAUNCH(RPT, F64, BBOX, SH) hipLaunchKernelGGL((k_filter<RPT, F64, BBOX, SH>), grid, block, 0, s, nd, m, t)
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

template <int MODEL, bool SINR>
static void launch_exact_m(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg)
{
    const dim3 grid(4, kShards), block(256);
    const int seg = t.use_matrix ? 0 : scan_variant(t.n_cnt);
#define RM_EX(ST, SG) hipLaunchKernelGGL((k_exact<MODEL, SINR, ST, SG>), grid, block, 0, s, nd, m, t)
    if (cfg.stochastic) {
        if (seg == 0) RM_EX(true, 0); else if (seg == 1) RM_EX(true, 1); else if (seg == 3) RM_EX(true, 3); else if (seg == 4) RM_EX(true, 4); else RM_EX(true, 2);
    } else {
        if (seg == 0) RM_EX(false, 0); else if (seg == 1) RM_EX(false, 1); else if (seg == 3) RM_EX(false, 3); else if (seg == 4) RM_EX(false, 4); else RM_EX(false, 2);
    }
#undef RM_EX
}

hipError_t launch_exact(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg)
{
    switch (m.kind) {
    case RM_MODEL_NULL: launch_exact_m<RM_MODEL_NULL, false>(s, nd, m, t, cfg); break;
    case RM_MODEL_UDGM: launch_exact_m<RM_MODEL_UDGM, false>(s, nd, m, t, cfg); break;
    case RM_MODEL_UDGM_CONST: launch_exact_m<RM_MODEL_UDGM_CONST, false>(s, nd, m, t, cfg); break;
    case RM_MODEL_N2N: launch_exact_m<RM_MODEL_N2N, false>(s, nd, m, t, cfg); break;
    case RM_MODEL_LOGDIST:
        if (m.flags & RM_LD_SINR) launch_exact_m<RM_MODEL_LOGDIST, true>(s, nd, m, t, cfg);
        else launch_exact_m<RM_MODEL_LOGDIST, false>(s, nd, m, t, cfg);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// sorted tables with more frames than the fused scans hold: seg_off before k_exact
hipError_t launch_seg_scan(hipStream_t s, const TickDev &t)
{
    if (!t.use_matrix && t.n_cnt > kFusedScanMax)
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, t.cand_tot, t.seg_off, t.n_cnt);
    return hipGetLastError();
}

hipError_t launch_self_entries(hipStream_t s, const NodesDev &nd, const TickDev &t)
{
    const int n_eval = t.n_active - t.first_eval;
    if (n_eval <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_self_entries, dim3(cdiv(n_eval, 256)), dim3(256), 0, s, nd, t);
    return hipGetLastError();
}

// unsorted tables: cell offsets + frame scan; sorted tables with very many frames: slot_off
hipError_t launch_offsets(hipStream_t s, const TickDev &t)
{
    if (t.use_matrix) {
        if (t.n_cnt > 0 && t.n_slabs > 0) hipLaunchKernelGGL(k_cell_off, dim3(t.n_cnt / 64), dim3(1024), 0, s, t);
        hipLaunchKernelGGL(k_slot_scan, dim3(1), dim3(1024), 0, s, t);
    } else if (t.n_cnt > kFusedScanMax) {
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, t.cursor, t.slot_off, t.n_cnt);
    }
    return hipGetLastError();
}

hipError_t launch_sinr(hipStream_t s, const ModelDev &m, const TickDev &t)
{
    hipLaunchKernelGGL(k_sinr, dim3(4, kShards), dim3(256), 0, s, m, t);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_pkt_interference(ModelDev m, TickDev t)
{
    write_pkt_interference(m, t, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// unsorted tables only
hipError_t launch_finalize(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                           const LaunchCfg &cfg)
{
    (void)nd;
    const dim3 grid(4, kShards), block(256);
    if (cfg.stochastic) {
        hipLaunchKernelGGL(k_finalize<true>, grid, block, 0, s, m, t);
    } else {
        hipLaunchKernelGGL(k_finalize<false>, grid, block, 0, s, m, t);
        hipLaunchKernelGGL(k_pkt_interference, dim3(8), dim3(256), 0, s, m, t);
    }
    return hipGetLastError();
}

// sorted tables only
hipError_t launch_reorder(hipStream_t s, const ModelDev &m, const TickDev &t, const LaunchCfg &cfg)
{
    const int n_new = t.n_active - t.first_new;
    const dim3 grid(max(1, min(2048, (n_new + 3) / 4))), block(256);
    const bool sinr = (m.kind == RM_MODEL_LOGDIST) && (m.flags & RM_LD_SINR);
    const int mode = scan_variant(t.n_cnt);
#define RM_RE(ST, SI, MO) hipLaunchKernelGGL((k_reorder<ST, SI, MO>), grid, block, 0, s, m, t)
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

// rm_batch_*: n independent ticks (sorted table, fp32 frame, no SINR, <= kFusedScanMax frames each)
// in four launches (+ k_store_ticks).  stage 0: k_tick_prep + k_filter_wg, 1: k_exact, 2: k_reorder.
hipError_t launch_pack_tx_batch(hipStream_t s, const NodesDev &nd, const int32_t *dev_src, int n_ticks, int n,
                                const int64_t *start_us, int64_t air_us, rm_tx_record *out)
{
    if (n <= 0 || n_ticks <= 0) return hipSuccess;
    if (n_ticks > kMaxBatch) return hipErrorInvalidValue;
    PackStarts st{};
    for (int b = 0; b < n_ticks; ++b) st.start_us[b] = start_us[b];
    hipLaunchKernelGGL(k_pack_tx_batch, dim3(cdiv(n, 256), n_ticks), dim3(256), 0, s, nd, dev_src, n, st, air_us, out);
    return hipGetLastError();
}

hipError_t launch_store_record(hipStream_t s, const rm_tx_record &r, rm_tx_record *dst)
{
    hipLaunchKernelGGL(k_store_record, dim3(1), dim3(64), 0, s, r, dst);
    return hipGetLastError();
}

hipError_t launch_transmit_one(hipStream_t s, const NodesDev &nd, const ModelDev &m, const rm_tx_record &tx,
                               uint64_t *rng_state, TransmitResult *host_mapped, uint32_t seq)
{
    const dim3 grid(1), block(1024);
    switch (m.kind) {
    case RM_MODEL_UDGM: hipLaunchKernelGGL(k_transmit_one<RM_MODEL_UDGM>, grid, block, 0, s, nd, m, tx, rng_state, host_mapped, seq); break;
    case RM_MODEL_UDGM_CONST:
        hipLaunchKernelGGL(k_transmit_one<RM_MODEL_UDGM_CONST>, grid, block, 0, s, nd, m, tx, rng_state, host_mapped, seq);
        break;
    case RM_MODEL_LOGDIST: hipLaunchKernelGGL(k_transmit_one<RM_MODEL_LOGDIST>, grid, block, 0, s, nd, m, tx, rng_state, host_mapped, seq); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_pack_result(hipStream_t s, const TickDev &t, TransmitResult *host_mapped)
{
    hipLaunchKernelGGL(k_pack_result, dim3(4), dim3(256), 0, s, t, host_mapped);
    return hipGetLastError();
}

bool batch_eligible(const TickDev &t, const LaunchCfg &cfg, const ModelDev &m)
{
    const bool sinr = (m.kind == RM_MODEL_LOGDIST) && (m.flags & RM_LD_SINR);
    return cfg.sorted && cfg.bbox && !cfg.f64_filter && !sinr && !t.use_matrix && t.n_cnt <= kFusedScanMax &&
           t.filter_mode == kFilterWg && t.n_active > t.first_new && t.n_rx > 0;
}

hipError_t launch_store_ticks(hipStream_t s, const TickDev *ticks, int n, TickDev *dev_ticks)
{
    for (int b0 = 0; b0 < n; b0 += kStoreTicks) {
        TickGroup g{};
        const int k = min(kStoreTicks, n - b0);
        for (int i = 0; i < k; ++i) g.t[i] = ticks[b0 + i];
        hipLaunchKernelGGL(k_store_ticks, dim3(k), dim3(64), 0, s, g, dev_ticks + b0, k);
    }
    return hipGetLastError();
}

// `ticks`: the host copies (grid sizes), `b`: the same descriptors in device memory
hipError_t launch_batch_stage(hipStream_t s, int stage, const NodesDev &nd, const ModelDev &m, const TickDev *ticks, int n,
                              const TickDev *b, const LaunchCfg &cfg)
{
    if (n < 1 || n > kMaxBatch) return hipErrorInvalidValue;
    int max_eval = 0, max_new = 0;
    int scan = 3; // the smallest fused-scan variant that holds every tick's counts
    for (int i = 0; i < n; ++i) {
        const int v = scan_variant(ticks[i].n_cnt);
        scan = (v == 1 || scan == 1) ? 1 : max(scan, v);
        max_eval = max(max_eval, ticks[i].n_active - ticks[i].first_eval);
        max_new = max(max_new, ticks[i].n_active - ticks[i].first_new);
    }
    const TickDev &t0 = ticks[0];
    if (stage == 0) {
        hipLaunchKernelGGL(k_tick_prep_batch, dim3(cdiv(max_eval, 256), 1, n), dim3(256), 0, s, nd, m, b);
        const dim3 grid(cdiv(t0.n_slabs, kWavesPerBlock), 1, n), block(kBlock);
        if (t0.rpt == 4) {
            if (cfg.shadow) hipLaunchKernelGGL((k_filter_wg_batch<4, true>), grid, block, 0, s, nd, m, b);
            else hipLaunchKernelGGL((k_filter_wg_batch<4, false>), grid, block, 0, s, nd, m, b);
        } else if (t0.rpt == 2) {
            if (cfg.shadow) hipLaunchKernelGGL((k_filter_wg_batch<2, true>), grid, block, 0, s, nd, m, b);
            else hipLaunchKernelGGL((k_filter_wg_batch<2, false>), grid, block, 0, s, nd, m, b);
        } else {
            if (cfg.shadow) hipLaunchKernelGGL((k_filter_wg_batch<1, true>), grid, block, 0, s, nd, m, b);
            else hipLaunchKernelGGL((k_filter_wg_batch<1, false>), grid, block, 0, s, nd, m, b);
        }
    } else if (stage == 1) {
        const dim3 grid(kShards, 1, n), block(256); // packed: one started chunk of 256 entries per workgroup at the bench sizes
#define RM_EXB(MODEL)                                                                                                \
    do {                                                                                                             \
        if (cfg.stochastic) {                                                                                        \
            if (scan == 3) hipLaunchKernelGGL((k_exact_batch<MODEL, true, 3>), grid, block, 0, s, nd, m, b);         \
            else if (scan == 4) hipLaunchKernelGGL((k_exact_batch<MODEL, true, 4>), grid, block, 0, s, nd, m, b);    \
            else hipLaunchKernelGGL((k_exact_batch<MODEL, true, 1>), grid, block, 0, s, nd, m, b);                   \
        } else {                                                                                                     \
            if (scan == 3) hipLaunchKernelGGL((k_exact_batch<MODEL, false, 3>), grid, block, 0, s, nd, m, b);        \
            else if (scan == 4) hipLaunchKernelGGL((k_exact_batch<MODEL, false, 4>), grid, block, 0, s, nd, m, b);   \
            else hipLaunchKernelGGL((k_exact_batch<MODEL, false, 1>), grid, block, 0, s, nd, m, b);                  \
        }                                                                                                            \
    } while (0)
        switch (m.kind) {
        case RM_MODEL_NULL: RM_EXB(RM_MODEL_NULL); break;
        case RM_MODEL_UDGM: RM_EXB(RM_MODEL_UDGM); break;
        case RM_MODEL_UDGM_CONST: RM_EXB(RM_MODEL_UDGM_CONST); break;
        case RM_MODEL_N2N: RM_EXB(RM_MODEL_N2N); break;
        case RM_MODEL_LOGDIST: RM_EXB(RM_MODEL_LOGDIST); break;
        default: return hipErrorInvalidValue;
        }
#undef RM_EXB
    } else {
        // two frames per wave (every workgroup redoes the scan of the per-frame counts first: fewer, longer
        // workgroups); a receiver partition hears 1/share of a frame's links, so its waves take more
        const int fpw = max(2, min(8, nd.n_rx > 0 ? nd.n / nd.n_rx : 1));
        const dim3 grid(max(1, min(2048, cdiv(max_new, 4 * fpw))), 1, n), block(256);
        if (cfg.stochastic) {
            if (scan == 3) hipLaunchKernelGGL((k_reorder_batch<true, 3>), grid, block, 0, s, m, b);
            else if (scan == 4) hipLaunchKernelGGL((k_reorder_batch<true, 4>), grid, block, 0, s, m, b);
            else hipLaunchKernelGGL((k_reorder_batch<true, 1>), grid, block, 0, s, m, b);
        } else {
            if (scan == 3) hipLaunchKernelGGL((k_reorder_batch<false, 3>), grid, block, 0, s, m, b);
            else if (scan == 4) hipLaunchKernelGGL((k_reorder_batch<false, 4>), grid, block, 0, s, m, b);
            else hipLaunchKernelGGL((k_reorder_batch<false, 1>), grid, block, 0, s, m, b);
        }
    }
    return hipGetLastError();
}

// draws, part 1: which ordered records need a draw, and how many per packet
hipError_t launch_draws_scan(hipStream_t s, const TickDev &t)
{
    const int tiles = cdiv(int(t.cap), kScanTile);
    const int n_new = t.n_active - t.first_new;
    hipLaunchKernelGGL(k_draw_tile_sums, dim3(tiles), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_draw_tile_scan, dim3(1), dim3(1024), 0, s, t);
    hipLaunchKernelGGL(k_draw_scan, dim3(tiles), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_pkt_draw_counts, dim3(max(1, cdiv(n_new, 256))), dim3(256), 0, s, t);
    return hipGetLastError();
}

// draws, part 2: walk the generator over the packets, then every flagged record draws at its place
hipError_t launch_draws_apply(hipStream_t s, const ModelDev &m, const TickDev &t, const uint32_t *all_cnt, int world,
                              int rank)
{
    hipLaunchKernelGGL(k_rng_chain, dim3(1), dim3(1024), 0, s, m, t, all_cnt, world, rank);
    hipLaunchKernelGGL(k_apply_draws, dim3(1024), dim3(256), 0, s, t);
    return hipGetLastError();
}

} // namespace rm
