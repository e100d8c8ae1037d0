// rm_math.hpp -- the exact-arithmetic layer: E-math of the extension spec, link hash, Q80 fixed point, java.util.Random
// (part of libradiomedium_hip.so; gfx950 only, -ffp-contract=off, no fast-math; overview at the top of rm_engine.h)
#pragma once

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
    // Branch-free: the central expression and the tail expression are both evaluated and one is selected --
    // operation for operation what the extension spec writes as three cases (the upper tail is the lower
    // tail's expression on 1 - u with the sign flipped).  A wave almost always holds lanes of the centre AND
    // of a tail (4.85 % of the deviates are tail values), so divergent branches would be issued one after
    // the other anyway; as one basic block the two dependent chains (and the caller's distance logarithm)
    // interleave, which is what a latency-bound evaluation -- one frame per workgroup, rm_tick.hip -- needs.
    const bool lower = u < 0.02425;
    const bool tail = lower || !(u <= 0.97575);
    const double t = tail ? (lower ? u : 1.0 - u) : 0.5; // (any value with a finite logarithm for the lanes that discard it)
    const double qt = sqrt(-2.0 * (det_log2(t) * 0.6931471805599453));
    const double rt = (((((c1 * qt + c2) * qt + c3) * qt + c4) * qt + c5) * qt + c6) /
                      ((((d1 * qt + d2) * qt + d3) * qt + d4) * qt + 1.0);
    const double q = u - 0.5;
    const double r = q * q;
    const double rc = (((((a1 * r + a2) * r + a3) * r + a4) * r + a5) * r + a6) * q /
                      (((((b1 * r + b2) * r + b3) * r + b4) * r + b5) * r + 1.0);
    return tail ? (lower ? rt : -rt) : rc;
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

RM_HD U128 u128_sub(U128 a, U128 b) // a >= b
{
    U128 r;
    r.lo = a.lo - b.lo;
    r.hi = a.hi - b.hi - (a.lo < b.lo ? 1u : 0u);
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


RM_HD double lcg_next_double(uint64_t &s)
{
    s = (s * kLcgA + kLcgC) & kLcgMask;
    const int64_t hi = int64_t(s >> 22); // next(26)
    s = (s * kLcgA + kLcgC) & kLcgMask;
    const int64_t lo = int64_t(s >> 21); // next(27)
    return double((hi << 27) + lo) * 0x1.0p-53;
}

} // namespace rm
