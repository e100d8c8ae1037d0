// rm_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the radio-medium engine.
//
// One heavy kernel, k_allpairs, sweeps every (frame on the air) x (receiver of this rank's
// partition) link: one receiver per lane (RPT receivers per thread in registers), transmitter
// tiles of 64 frames staged in LDS, a conservative fp32 (or fp64) geometric pre-filter whose
// wave ballot decides whether the exact fp64 evaluation -- the reference's arithmetic, in the
// reference's operation order -- has to run for that (frame, 64 receivers) step.  Heard links
// are appended to an unordered staging list together with their rank inside the
// (frame, receiver-slab) cell; everything after that is O(heard links): offsets from the cell
// counts, SINR over per-receiver lists, ordered scatter, Java-RNG draws.
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (no fast-math): the exact path relies
// on every fp64 operation being one IEEE-754 rounding, as in Java.
//
// Reference paths: /root/reference/radio-medium/java/se/sics/emul8/radiomedium/.

#include "rm_engine.h"

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
    if (u < 0.02425) {
        const double q = sqrt(-2.0 * (det_log2(u) * 0.6931471805599453));
        return (((((c1 * q + c2) * q + c3) * q + c4) * q + c5) * q + c6) /
               ((((d1 * q + d2) * q + d3) * q + d4) * q + 1.0);
    }
    if (u <= 0.97575) {
        const double q = u - 0.5;
        const double r = q * q;
        return (((((a1 * r + a2) * r + a3) * r + a4) * r + a5) * r + a6) * q /
               (((((b1 * r + b2) * r + b3) * r + b4) * r + b5) * r + 1.0);
    }
    const double q = sqrt(-2.0 * (det_log2(1.0 - u) * 0.6931471805599453));
    return -((((((c1 * q + c2) * q + c3) * q + c4) * q + c5) * q + c6) /
             ((((d1 * q + d2) * q + d3) * q + d4) * q + 1.0));
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

// per-node pre-filter record: (fx, fy, fz, channel bits); a disabled radio gets a NaN position so
// that the geometric test can never pass (Transciever.isEnabled(), UDGMRadioMedium.java:102)
__global__ void __launch_bounds__(256) k_prep_rx(NodesDev nd, ModelDev m)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd.n) return;
    float4 r;
    const bool geometric = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_UDGM_CONST || m.kind == RM_MODEL_LOGDIST);
    if (!nd.enabled[i]) {
        r.x = r.y = r.z = __builtin_nanf("");
    } else if (geometric) {
        r.x = float(nd.x[i] - m.org_x);
        r.y = float(nd.y[i] - m.org_y);
        r.z = float(nd.z[i] - m.org_z);
    } else {
        r.x = r.y = r.z = 0.f;
    }
    r.w = __int_as_float(nd.channel[i]);
    nd.rxf[i] = r;
}

RM_D float round_up_to_float(double v)
{
    float f = float(v);
    if (double(f) < v) f = nextafterf(f, __builtin_inff());
    return f;
}

// per-frame pre-filter record: (fx, fy, fz, threshold on the fp32 squared distance)
__global__ void __launch_bounds__(256) k_prep_tx(ModelDev m, TickDev t)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // eval-relative
    const int n_eval = t.n_active - t.first_eval;
    if (i >= n_eval) return;
    const rm_tx_record tx = t.tx[t.first_eval + i];
    double cut; // cut-off distance (metres): no link beyond it can matter; <0 nobody, inf everybody
    const double inf = u2f(0x7FF0000000000000ull);
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
            cut = m.ld_d0 * exp2(margin / (10.0 * m.ld_exp) * 3.3219280948873622) * (1.0 + 1e-6);
            if (cut < m.ld_d0) cut = m.ld_d0;
        }
    } else {
        cut = inf; // Null / N2N: no geometry
    }
    float4 f;
    double thr64;
    const double rx_ = tx.x - m.org_x, ry_ = tx.y - m.org_y, rz_ = tx.z - m.org_z;
    const bool geometric = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_UDGM_CONST || m.kind == RM_MODEL_LOGDIST);
    const bool in_frame = fabs(rx_) <= m.coord_bound && fabs(ry_) <= m.coord_bound && fabs(rz_) <= m.coord_bound;
    if (cut < 0.0) {
        f.x = f.y = f.z = 0.f;
        f.w = -1.f;
        thr64 = -1.0;
    } else if (!geometric || !in_frame || cut == inf) {
        // everything (enabled, same channel) is a candidate; the exact path decides
        f.x = f.y = f.z = 0.f;
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
    t.txf[i] = f;
    t.txd[i] = thr64;
}

// RadioPacket(node, time, data): copies the source radio's txpower / channel (RadioPacket.java:46-52)
__global__ void __launch_bounds__(256)
k_pack_tx(NodesDev nd, const int32_t *src, int n, int64_t start_us, int64_t air_us, rm_tx_record *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = src[i];
    rm_tx_record r;
    if (s < 0 || s >= nd.n) {
        r.x = r.y = r.z = 0.0;
        r.txpower = 0.0;
        r.txprob = 0.0;
        r.start_us = start_us;
        r.air_us = 0;
        r.src = -1;
        r.channel = 0;
    } else {
        r.x = nd.x[s];
        r.y = nd.y[s];
        r.z = nd.z[s];
        r.txpower = nd.txpower[s];
        r.txprob = nd.txprob[s];
        r.start_us = start_us;
        r.air_us = air_us;
        r.src = s;
        r.channel = nd.channel[s];
    }
    out[i] = r;
}

// ============================================================================ the all-pairs kernel

struct LinkEval {
    bool append;   // goes to the staging list
    bool wanted;   // heard link of a new frame: gets an output record
    uint8_t flags;
    double aux;    // probability (UDGM / N2N) or rssi (logdist)
    double lin;    // linear power (SINR)
};

// Exact evaluation of one link, in the reference's order of tests
// (UDGMRadioMedium.java:99-111, N2NRadioMedium.java:55-67, NullRadioMedium.java:62-73,
//  UDGMConstantLossRadioMedium.java:25-33).
template <int MODEL, bool SINR>
RM_D LinkEval eval_link(const ModelDev &m, const NodesDev &nd, const rm_tx_record &tx, int j, bool is_new)
{
    LinkEval r;
    r.append = false;
    r.wanted = false;
    r.flags = 0;
    r.aux = 0.0;
    r.lin = 0.0;
    if (j == tx.src) return r;                  // node != source
    if (!nd.enabled[j]) return r;               // radio.isEnabled()
    if (nd.channel[j] != tx.channel) return r;  // radio.getWirelessChannel() == channel
    if (MODEL == RM_MODEL_NULL) {
        r.append = r.wanted = true;
        r.flags = kFlagHeardNew;
        return r;
    }
    if (MODEL == RM_MODEL_N2N) {
        // N2NRadioMedium.java:28-37
        const int sid = nd.int_id[tx.src];
        const int did = nd.int_id[j];
        double p = 0.0;
        if (m.n2n != nullptr && sid > 0 && did > 0 && sid <= m.n2n_m && did <= m.n2n_m) {
            p = m.n2n[int64_t(sid - 1) * m.n2n_m + (did - 1)] * nd.rxprob[j];
        }
        if (p <= 0.0) return r;
        r.append = r.wanted = true;
        r.flags = kFlagHeardNew;
        r.aux = p;
        return r;
    }
    const double rx = nd.x[j], ry = nd.y[j], rz = nd.z[j];
    if (MODEL == RM_MODEL_UDGM_CONST) {
        const double d = ref_distance(tx.x, tx.y, tx.z, rx, ry, rz);
        if (d < m.const_range) {
            r.append = r.wanted = true;
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
        const double p = ratio * nd.rxprob[j];
        if (p <= 0.0) return r;
        r.append = r.wanted = true;
        r.flags = kFlagHeardNew;
        r.aux = p;
        return r;
    }
    if (MODEL == RM_MODEL_LOGDIST) {
        const double rssi = logdist_rssi(m, tx, rx, ry, rz, j);
        const bool heard = is_new && (rssi >= m.ld_sens) && !(nd.rxprob[j] <= 0.0);
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

RM_D uint32_t lane_prefix(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

template <int MODEL, int RPT, bool F64, bool SINR>
__global__ void __launch_bounds__(kBlock) k_allpairs(const NodesDev nd, const ModelDev m, const TickDev t)
{
    __shared__ float4 s_txf[kTxChunk];
    __shared__ double s_thr[kTxChunk];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slab = blockIdx.x * kWavesPerBlock + wave;
    const int chunk = blockIdx.y;
    const int n_eval = t.n_active - t.first_eval;
    const int e0 = chunk * kTxChunk;                    // eval-relative index of the tile's first frame
    const int nt = min(kTxChunk, n_eval - e0);

    // stage the transmitter tile (coalesced) -- the north-star's "LDS-staged transmitter tiles"
    if (threadIdx.x < nt) {
        s_txf[threadIdx.x] = t.txf[e0 + threadIdx.x];
        if (F64) s_thr[threadIdx.x] = t.txd[e0 + threadIdx.x];
    }
    __syncthreads();
    if (slab >= t.n_slabs) return;

    const int rx_end = t.rx_first + t.rx_count;
    const int jbase = t.rx_first + slab * (64 * RPT);

    // receivers of this lane, resident in registers for the whole tile
    float fx[RPT], fy[RPT], fz[RPT];
    int fch[RPT];
    double gx[RPT], gy[RPT], gz[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int j = jbase + r * 64 + lane;
        if (j < rx_end) {
            const float4 v = nd.rxf[j];
            fx[r] = v.x;
            fy[r] = v.y;
            fz[r] = v.z;
            fch[r] = __float_as_int(v.w);
            if (F64) {
                gx[r] = nd.x[j];
                gy[r] = nd.y[j];
                gz[r] = nd.z[j];
                if (v.x != v.x) gx[r] = u2f(0x7FF8000000000000ull); // disabled
            }
        } else {
            fx[r] = fy[r] = fz[r] = __builtin_nanf("");
            fch[r] = 0;
            if (F64) gx[r] = gy[r] = gz[r] = u2f(0x7FF8000000000000ull);
        }
    }

    uint32_t cnt_lane = 0; // lane l ends up holding the heard count of the tile's l-th frame
    const rm_tx_record *txg = t.tx + t.first_eval + e0;

    for (int ti = 0; ti < nt; ++ti) {
        const float4 tf = s_txf[ti];
        const int tch = txg[ti].channel; // wave-uniform: scalar load
        uint64_t mask[RPT];
        uint64_t any = 0;
        if (F64) {
            const double tx_ = txg[ti].x, ty_ = txg[ti].y, tz_ = txg[ti].z;
            const double thr = s_thr[ti];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const double dx = tx_ - gx[r], dy = ty_ - gy[r], dz = tz_ - gz[r];
                const double s = dx * dx + dy * dy + dz * dz;
                mask[r] = __ballot((s <= thr) && (fch[r] == tch));
                any |= mask[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const float dx = fx[r] - tf.x, dy = fy[r] - tf.y, dz = fz[r] - tf.z;
                const float s = dx * dx + dy * dy + dz * dz;
                mask[r] = __ballot((s <= tf.w) && (fch[r] == tch));
                any |= mask[r];
            }
        }
        if (any == 0) continue; // the common case: nobody of these 64*RPT receivers is near this frame

        // ---- exact path (rare): the reference's arithmetic on the candidate lanes
        const rm_tx_record tx = txg[ti];
        const int e_abs = e0 + ti;
        const bool is_new = (t.first_eval + e_abs) >= t.first_new;
        uint32_t run = 0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (mask[r] == 0) continue; // wave-uniform
            const int j = jbase + r * 64 + lane;
            LinkEval ev;
            ev.append = ev.wanted = false;
            ev.flags = 0;
            ev.aux = ev.lin = 0.0;
            if ((mask[r] >> lane) & 1ull) ev = eval_link<MODEL, SINR>(m, nd, tx, j, is_new);
            const uint64_t ab = __ballot(ev.append);
            if (ab == 0) continue;
            const uint64_t wb = __ballot(ev.wanted);
            // one atomic per wave step: the first appending lane reserves the block of entries
            uint32_t base = 0;
            const int leader = __ffsll((long long)ab) - 1;
            if (lane == leader) base = atomicAdd(t.stage_count, uint32_t(__popcll(ab)));
            base = __shfl(base, leader);
            if (ev.append) {
                const uint32_t idx = base + lane_prefix(ab);
                if (idx < t.cap) {
                    t.st_pkt[idx] = e_abs;
                    t.st_dst[idx] = j;
                    t.st_rank[idx] = run + lane_prefix(wb);
                    t.st_flags[idx] = ev.flags;
                    if (MODEL == RM_MODEL_UDGM || MODEL == RM_MODEL_N2N || MODEL == RM_MODEL_LOGDIST)
                        t.st_aux[idx] = ev.aux;
                    if (SINR) {
                        t.st_lin[idx] = ev.lin;
                        t.st_next[idx] = atomicExch(&t.head[j - t.rx_first], int(idx));
                    }
                } else {
                    t.stage_count[1] = 1u;
                }
            }
            run += uint32_t(__popcll(wb));
        }
        if (lane == ti) cnt_lane = run;
    }

    // heard counts of this (tile, slab) cell: one coalesced 256-byte store per wave
    const int cc = (e0 - t.cnt_base) / kTxChunk;
    if (cc >= 0) t.cnt[(size_t(cc) * t.n_slabs + slab) * 64 + lane] = cnt_lane;
}

// half duplex (SINR mode): every frame on the air leaves a SELF entry in its source's list
__global__ void __launch_bounds__(256) k_self_entries(TickDev t)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_eval = t.n_active - t.first_eval;
    if (e >= n_eval) return;
    const int src = t.tx[t.first_eval + e].src;
    if (src < t.rx_first || src >= t.rx_first + t.rx_count) return;
    const uint32_t idx = atomicAdd(t.stage_count, 1u);
    if (idx >= t.cap) {
        t.stage_count[1] = 1u;
        return;
    }
    t.st_pkt[idx] = e;
    t.st_dst[idx] = src;
    t.st_rank[idx] = 0;
    t.st_flags[idx] = kFlagSelf;
    t.st_aux[idx] = 0.0;
    t.st_lin[idx] = 0.0;
    t.st_next[idx] = atomicExch(&t.head[src - t.rx_first], int(idx));
}

// ============================================================================ offsets (tiny scans)

// partial[g][slot] = sum of cnt over the slabs of group g
__global__ void __launch_bounds__(256) k_partial(TickDev t)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= t.n_cnt * t.n_groups) return;
    const int slot = idx % t.n_cnt;
    const int g = idx / t.n_cnt;
    const int cc = slot >> 6, tl = slot & 63;
    const int s0 = g * t.slabs_per_group;
    const int s1 = min(t.n_slabs, s0 + t.slabs_per_group);
    uint32_t sum = 0;
    for (int s = s0; s < s1; ++s) sum += t.cnt[(size_t(cc) * t.n_slabs + s) * 64 + tl];
    t.partial[size_t(g) * t.n_cnt + slot] = sum;
}

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

// slot_off = exclusive scan over slots of the per-slot totals; also publishes the heard-link count
__global__ void __launch_bounds__(1024) k_slot_scan(TickDev t)
{
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (int base = 0; base < t.n_cnt; base += 1024) {
        const int slot = base + threadIdx.x;
        uint32_t v = 0;
        if (slot < t.n_cnt) {
            for (int g = 0; g < t.n_groups; ++g) v += t.partial[size_t(g) * t.n_cnt + slot];
        }
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
        if (slot < t.n_cnt) t.slot_off[slot] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        t.slot_off[t.n_cnt] = carry;
        t.out_count[0] = carry < t.cap ? carry : t.cap;
        t.out_count[1] = (carry > t.cap || t.stage_count[1] != 0u) ? 1u : 0u;
        t.out_count[2] = carry;
    }
}

// off[cell] = first output position of the (slot, slab) cell
__global__ void __launch_bounds__(256) k_slab_off(TickDev t)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= t.n_cnt * t.n_groups) return;
    const int slot = idx % t.n_cnt;
    const int g = idx / t.n_cnt;
    const int cc = slot >> 6, tl = slot & 63;
    uint32_t run = t.slot_off[slot];
    for (int gg = 0; gg < g; ++gg) run += t.partial[size_t(gg) * t.n_cnt + slot];
    const int s0 = g * t.slabs_per_group;
    const int s1 = min(t.n_slabs, s0 + t.slabs_per_group);
    for (int s = s0; s < s1; ++s) {
        const size_t c = (size_t(cc) * t.n_slabs + s) * 64 + tl;
        t.off[c] = run;
        run += t.cnt[c];
    }
}

// ============================================================================ SINR (O(heard links))

RM_D bool frames_overlap(const rm_tx_record &w, const rm_tx_record &k)
{
    return k.start_us < w.start_us + w.air_us && k.start_us + k.air_us > w.start_us;
}

// one thread per staging entry that is a heard link of a new frame: walk the receiver's list,
// sum the co-channel, time-overlapping interferers exactly (Q80), apply capture + half duplex
__global__ void __launch_bounds__(256) k_sinr(ModelDev m, TickDev t)
{
    const uint32_t n = min(t.stage_count[0], t.cap);
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        if (!(t.st_flags[e] & kFlagHeardNew)) continue;
        const int j = t.st_dst[e];
        const rm_tx_record w = t.tx[t.first_eval + t.st_pkt[e]];
        U128 acc = {0, 0};
        bool half_duplex = false;
        for (int idx = t.head[j - t.rx_first]; idx >= 0; idx = t.st_next[idx]) {
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

RM_D double tx_success(const ModelDev &m, const rm_tx_record &tx)
{
    // UDGMRadioMedium.java:63-65 uses successRatioRx (sic); N2NRadioMedium.java:24-26
    if (m.kind == RM_MODEL_UDGM) return m.udgm_ratio_rx * tx.txprob;
    return tx.txprob;
}

// staging entry -> final position off[cell] + rank ; verdict for everything that needs no draw
template <bool STOCH>
__global__ void __launch_bounds__(256) k_finalize_impl(ModelDev m, TickDev t, const double *rxprob)
{
    const uint32_t n = min(t.stage_count[0], t.cap);
    const bool sinr = (m.kind == RM_MODEL_LOGDIST) && (m.flags & RM_LD_SINR);
    const bool draws_possible = (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_N2N || m.kind == RM_MODEL_LOGDIST);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        if (!(t.st_flags[e] & kFlagHeardNew)) continue;
        const int erel = t.st_pkt[e];
        const int j = t.st_dst[e];
        const int slot = erel - t.cnt_base;
        const int slab = (j - t.rx_first) / (64 * t.rpt);
        const uint32_t pos = t.off[(size_t(slot >> 6) * t.n_slabs + slab) * 64 + (slot & 63)] + t.st_rank[e];
        if (pos >= t.cap) continue;
        const rm_tx_record tx = t.tx[t.first_eval + erel];
        t.out_pkt[pos] = slot - t.shift;
        t.out_dst[pos] = j;
        double rssi = tx.txpower; // reference models hand the packet's transmit power through
        double prob = 1.0;
        if (m.kind == RM_MODEL_LOGDIST) {
            rssi = t.st_aux[e];
            prob = rxprob[j];
        } else if (m.kind == RM_MODEL_UDGM || m.kind == RM_MODEL_N2N) {
            prob = t.st_aux[e];
        }
        t.out_rssi[pos] = rssi;
        t.out_sinr[pos] = sinr ? t.st_sinr[e] : 0.0;
        const bool collided = sinr && t.st_coll[e];
        if (STOCH) {
            t.out_prob[pos] = prob;
            t.out_verdict[pos] = collided ? uint8_t(RM_INTERFERED) : uint8_t(0); // 0 = pending
        } else {
            const bool interference = draws_possible && (tx_success(m, tx) <= 0.0);
            t.out_verdict[pos] = (interference || collided) ? uint8_t(RM_INTERFERED) : uint8_t(RM_DELIVERED);
        }
    }
    if (!STOCH) {
        const int n_new = t.n_active - t.first_new;
        for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < uint32_t(n_new); q += stride) {
            const rm_tx_record tx = t.tx[t.first_new + q];
            t.pkt_interference[q] = (draws_possible && tx_success(m, tx) <= 0.0) ? 1 : 0;
        }
    }
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

// The only sequential part: the shared generator is consumed packet after packet
// (Simulator.getRandom(); UDGMRadioMedium.java:85-92,106).  One workgroup: the per-packet jump
// maps are built in parallel, then one lane walks the packets.
__global__ void __launch_bounds__(1024) k_rng_chain(ModelDev m, TickDev t)
{
    __shared__ uint64_t s_A[1024], s_C[1024];
    __shared__ double s_txs[1024];
    __shared__ uint64_t s_state;
    const int n_new = t.n_active - t.first_new;
    const uint32_t n = t.out_count[0];
    if (threadIdx.x == 0) s_state = *t.rng_state & kLcgMask;
    __syncthreads();
    for (int base = 0; base < n_new; base += 1024) {
        const int q = base + threadIdx.x;
        if (q < n_new) {
            const uint32_t b = min(t.slot_off[q + t.shift], n);
            const uint32_t e = min(t.slot_off[q + t.shift + 1], n);
            const uint32_t draws = t.draw_scan[e] - t.draw_scan[b];
            uint64_t A, C;
            lcg_jump_map(2ull * draws, A, C);
            s_A[threadIdx.x] = A;
            s_C[threadIdx.x] = C;
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
                t.pkt_rng[base + i] = s;
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

// ============================================================================ launchers

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

hipError_t launch_prep_rx(hipStream_t s, const NodesDev &nd, const ModelDev &m)
{
    if (nd.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_prep_rx, dim3(cdiv(nd.n, 256)), dim3(256), 0, s, nd, m);
    return hipGetLastError();
}

hipError_t launch_prep_tx(hipStream_t s, const ModelDev &m, const TickDev &t)
{
    const int n_eval = t.n_active - t.first_eval;
    if (n_eval <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_prep_tx, dim3(cdiv(n_eval, 256)), dim3(256), 0, s, m, t);
    return hipGetLastError();
}

hipError_t launch_pack_tx(hipStream_t s, const NodesDev &nd, const int32_t *dev_src, int n, int64_t start_us,
                          int64_t air_us, rm_tx_record *out)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_tx, dim3(cdiv(n, 256)), dim3(256), 0, s, nd, dev_src, n, start_us, air_us, out);
    return hipGetLastError();
}

template <int MODEL, bool SINR>
static hipError_t launch_allpairs_m(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                                    const LaunchCfg &cfg)
{
    const int n_eval = t.n_active - t.first_eval;
    const dim3 grid(cdiv(t.n_slabs, kWavesPerBlock), cdiv(n_eval, kTxChunk));
    const dim3 block(kBlock);
#define RM_LAUNCH(RPT, F64) hipLaunchKernelGGL((k_allpairs<MODEL, RPT, F64, SINR>), grid, block, 0, s, nd, m, t)
    if (t.rpt == 4) {
        if (cfg.f64_filter) RM_LAUNCH(4, true); else RM_LAUNCH(4, false);
    } else {
        if (cfg.f64_filter) RM_LAUNCH(1, true); else RM_LAUNCH(1, false);
    }
#undef RM_LAUNCH
    return hipGetLastError();
}

hipError_t launch_allpairs(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                           const LaunchCfg &cfg)
{
    if (t.n_active - t.first_eval <= 0 || t.n_slabs <= 0) return hipSuccess;
    switch (m.kind) {
    case RM_MODEL_NULL: return launch_allpairs_m<RM_MODEL_NULL, false>(s, nd, m, t, cfg);
    case RM_MODEL_UDGM: return launch_allpairs_m<RM_MODEL_UDGM, false>(s, nd, m, t, cfg);
    case RM_MODEL_UDGM_CONST: return launch_allpairs_m<RM_MODEL_UDGM_CONST, false>(s, nd, m, t, cfg);
    case RM_MODEL_N2N: return launch_allpairs_m<RM_MODEL_N2N, false>(s, nd, m, t, cfg);
    case RM_MODEL_LOGDIST:
        if (m.flags & RM_LD_SINR) return launch_allpairs_m<RM_MODEL_LOGDIST, true>(s, nd, m, t, cfg);
        return launch_allpairs_m<RM_MODEL_LOGDIST, false>(s, nd, m, t, cfg);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_self_entries(hipStream_t s, const TickDev &t)
{
    const int n_eval = t.n_active - t.first_eval;
    if (n_eval <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_self_entries, dim3(cdiv(n_eval, 256)), dim3(256), 0, s, t);
    return hipGetLastError();
}

hipError_t launch_offsets(hipStream_t s, const TickDev &t)
{
    const int work = t.n_cnt * t.n_groups;
    if (work > 0) hipLaunchKernelGGL(k_partial, dim3(cdiv(work, 256)), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_slot_scan, dim3(1), dim3(1024), 0, s, t);
    if (work > 0) hipLaunchKernelGGL(k_slab_off, dim3(cdiv(work, 256)), dim3(256), 0, s, t);
    return hipGetLastError();
}

hipError_t launch_sinr(hipStream_t s, const ModelDev &m, const TickDev &t)
{
    hipLaunchKernelGGL(k_sinr, dim3(1024), dim3(256), 0, s, m, t);
    return hipGetLastError();
}

hipError_t launch_finalize(hipStream_t s, const NodesDev &nd, const ModelDev &m, const TickDev &t,
                           const LaunchCfg &cfg)
{
    if (cfg.stochastic)
        hipLaunchKernelGGL(k_finalize_impl<true>, dim3(1024), dim3(256), 0, s, m, t, nd.rxprob);
    else
        hipLaunchKernelGGL(k_finalize_impl<false>, dim3(1024), dim3(256), 0, s, m, t, nd.rxprob);
    return hipGetLastError();
}

hipError_t launch_draws(hipStream_t s, const ModelDev &m, const TickDev &t)
{
    const int tiles = cdiv(int(t.cap), kScanTile);
    hipLaunchKernelGGL(k_draw_tile_sums, dim3(tiles), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_draw_tile_scan, dim3(1), dim3(1024), 0, s, t);
    hipLaunchKernelGGL(k_draw_scan, dim3(tiles), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_rng_chain, dim3(1), dim3(1024), 0, s, m, t);
    hipLaunchKernelGGL(k_apply_draws, dim3(1024), dim3(256), 0, s, t);
    return hipGetLastError();
}

} // namespace rm
