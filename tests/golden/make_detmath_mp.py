#!/usr/bin/env python3
"""A second, structurally different evaluation of the extension spec's E-math, written from the
spec's text on a different arithmetic substrate: every operation is carried out EXACTLY on rationals
(fractions.Fraction) and then rounded once to binary64 (int / int true division in CPython is correctly rounded),
constants are derived with mpmath at 60 digits and rounded, integer functions use Python's unbounded ints.  Neither the
oracle (oracle/rm_oracle.c) nor the engine (radio-sim_amd/csrc/rm_math.hpp) is involved: both are then compared with
the fixture this script writes, so a transcription slip shared by the two C texts cannot hide.

    python tests/golden/make_detmath_mp.py        # rewrites tests/golden/detmath_mp.npz

It also checks, before writing anything, that the spec's polynomials approximate the true functions (mpmath) as well
as they claim: a wrong coefficient would fail here."""
import math
import os
from fractions import Fraction as F

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 60


def rn(q):
    """one IEEE-754 rounding to binary64 (nearest, ties to even) of an exact rational"""
    return float(q)


def add(a, b): return rn(F(a) + F(b))
def sub(a, b): return rn(F(a) - F(b))
def mul(a, b): return rn(F(a) * F(b))
def div(a, b): return rn(F(a) / F(b))


def sqrt(a):
    """correctly rounded square root: candidates around mpmath's 60-digit value, decided exactly"""
    if a == 0.0:
        return 0.0
    r = float(mp.sqrt(mp.mpf(a)))
    best = None
    for c in (math.nextafter(r, 0.0), r, math.nextafter(r, math.inf)):
        # |c - sqrt(a)| compared exactly through squares is awkward; mpmath at 60 digits decides safely
        err = abs(mp.mpf(c) - mp.sqrt(mp.mpf(a)))
        if best is None or err < best[0]:
            best = (err, c)
    return best[1]


# constants of the spec, derived -- and compared with the literals the spec prints
INV_LN2 = float(1 / mp.log(2))
LOG10_2 = float(mp.log10(2))
LOG2_10 = float(mp.log(10, 2))
SQRT2 = float(mp.sqrt(2))
LN2 = float(mp.log(2))
assert (LN2, INV_LN2, LOG10_2, LOG2_10, SQRT2) == (0.6931471805599453, 1.4426950408889634, 0.30102999566398120,
                                                   3.3219280948873622, 1.4142135623730951)

# Acklam's coefficients (the published table of the algorithm; accuracy against the true quantile is checked below)
A = [-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01,
     2.506628277459239e+00]
B = [-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01]
C = [-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00,
     2.938163982698783e+00]
D = [7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00]


def det_log2(x):
    m, e = math.frexp(x)              # x = m * 2^e, m in [0.5, 1)
    m, e = m * 2.0, e - 1             # exact: m in [1, 2)
    if m > SQRT2:
        m, e = mul(m, 0.5), e + 1
    f = div(sub(m, 1.0), add(m, 1.0))
    s = mul(f, f)
    q = div(1.0, 23.0)                # Horner of sum_{k=1..11} s^k / (2k+1), highest first, coefficients 1.0 / (2k+1)
    for k in range(10, 0, -1):
        q = add(mul(q, s), div(1.0, float(2 * k + 1)))
    q = mul(q, s)
    r = add(f, mul(f, q))
    return add(float(e), mul(mul(2.0, r), INV_LN2))


def det_exp2(y):
    if y != y:
        return y
    if not (y >= -1022.0):
        return 0.0
    if y > 1023.0:
        return math.inf
    k = float(math.floor(add(y, 0.5)))
    t = mul(sub(y, k), LN2)
    q = div(1.0, float(math.factorial(13)))   # Horner of sum_{n=0..13} t^n / n!, coefficients 1.0 / n!
    for n in range(12, -1, -1):
        q = add(mul(q, t), div(1.0, float(math.factorial(n))))
    return rn(F(q) * F(2) ** int(k))          # q * 2^k through the exponent bits: exact scaling (k >= -1022: no denormal)


def det_log10(x): return mul(det_log2(x), LOG10_2)
def det_pow10(y): return det_exp2(mul(y, LOG2_10))


def horner(cs, q, last=None):
    r = cs[0]
    for c in cs[1:]:
        r = add(mul(r, q), c)
    if last is not None:
        r = add(mul(r, q), last)
    return r


def det_normal(u):
    def tail(t):
        q = sqrt(mul(-2.0, mul(det_log2(t), LN2)))
        return div(horner(C, q), horner(D, q, 1.0))
    if u < 0.02425:
        return tail(u)
    if u <= 0.97575:
        q = sub(u, 0.5)
        r = mul(q, q)
        return div(mul(horner(A, r), q), horner(B, r, 1.0))
    return -tail(sub(1.0, u))


M64 = (1 << 64) - 1


def mix64(z):
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & M64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def link_hash(seed, a, b):
    lo, hi = min(a, b), max(a, b)
    return mix64(mix64((seed + 0x9E3779B97F4A7C15) & M64) ^ ((lo << 32) | hi))


def shadow_u(h):
    return rn((F(h >> 12) + F(1, 2)) * F(1, 2 ** 52))      # ((h >> 12) + 0.5) * 2^-52: exact in binary64


def q80_roundtrip(lin):
    """truncate to a multiple of 2^-80, back to the nearest-even double"""
    if not (lin > 0.0):
        return 0.0
    return rn(F(math.floor(F(lin) * (1 << 80)), 1 << 80))


def main():
    rng = np.random.default_rng(20260104)
    # accuracy of the spec's functions against the true ones (mpmath): a wrong constant or coefficient fails here
    for x in list(np.exp(rng.uniform(-60, 60, 60))) + [1.0, 2.0, 1024.0, SQRT2, math.nextafter(SQRT2, 2.0)]:
        assert abs(mp.mpf(det_log2(x)) - mp.log(mp.mpf(x), 2)) <= mp.mpf(4e-16) * max(1, abs(mp.log(mp.mpf(x), 2))), x
    for y in list(rng.uniform(-200, 200, 60)) + [0.0, 0.5, -0.5, 10.0]:
        assert abs(mp.mpf(det_exp2(y)) / mp.power(2, mp.mpf(y)) - 1) <= mp.mpf(5e-16), y
    for u in list(rng.uniform(0, 1, 80)) + [1e-12, 0.02425, 0.0242499, 0.97575, 0.975751, 1 - 1e-12, 0.5]:
        true = mp.sqrt(2) * mp.erfinv(2 * mp.mpf(u) - 1)
        assert abs(mp.mpf(det_normal(u)) - true) <= mp.mpf(1.2e-9) * max(1, abs(true)), u
    assert det_log2(1.0) == 0.0 and det_log2(1024.0) == 10.0 and det_exp2(10.0) == 1024.0 and det_normal(0.5) == 0.0

    xs = np.concatenate([np.exp(rng.uniform(-80, 80, 300)), [1.0, 2.0, 0.5, SQRT2, math.nextafter(SQRT2, 2.0), 1e-300, 1e300,
                                                               10.0, 50.0 / 1.0, 123.456]])
    ys = np.concatenate([rng.uniform(-300, 300, 250), rng.uniform(-12, 1, 100), [0.0, 0.5, -0.5, 1.5, -1022.0, 1023.0, -1100.0,
                                                                                   1100.0, -9.5, -10.0]])
    us = np.concatenate([rng.uniform(0, 1, 300), rng.uniform(0, 0.03, 60), rng.uniform(0.97, 1, 60),
                         [0.02425, math.nextafter(0.02425, 0.0), 0.97575, math.nextafter(0.97575, 1.0), 0.5, 2.0 ** -53,
                          1 - 2.0 ** -53]])
    lin = np.concatenate([10.0 ** rng.uniform(-28, 3, 200), [0.0, 2.0 ** -80, 2.0 ** -81, 1.0, 3 * 2.0 ** -80, 1e-10 * (1 + 2.0 ** -30)]])
    pairs = rng.integers(0, 2 ** 31 - 1, (300, 2)).astype(np.uint32)
    seed = 0xC0FFEE
    hs = np.array([link_hash(seed, int(a), int(b)) for a, b in pairs], dtype=np.uint64)
    np.savez_compressed(
        os.path.join(HERE, "detmath_mp.npz"),
        log2_x=xs, log2_y=np.array([det_log2(float(v)) for v in xs]),
        log10_y=np.array([det_log10(float(v)) for v in xs]),
        exp2_x=ys, exp2_y=np.array([det_exp2(float(v)) for v in ys]),
        pow10_y=np.array([det_pow10(float(v)) for v in ys / 10.0]),
        normal_u=us, normal_g=np.array([det_normal(float(v)) for v in us]),
        fixed_x=lin, fixed_y=np.array([q80_roundtrip(float(v)) for v in lin]),
        hash_pairs=pairs, hash_seed=np.array(seed, dtype=np.uint64), hash_h=hs,
        hash_u=np.array([shadow_u(int(h)) for h in hs]))
    print("detmath_mp ok: %d log2, %d exp2, %d normal, %d Q80, %d hash vectors" % (len(xs), len(ys), len(us), len(lin), len(pairs)))


if __name__ == "__main__":
    main()
