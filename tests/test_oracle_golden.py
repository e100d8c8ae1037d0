"""CPU tier: the oracle reproduces the committed golden vectors bit for bit, and its extension
math (DESIGN.md "Extension spec") is accurate against libm / scipy."""
import os

import numpy as np
import pytest

from golden_util import SCENARIOS, load, GOLDEN
from util import KINDS


def _nodes(O, g):
    nd = O.NodeTable(g["nodes"].n)
    for f in ("x", "y", "z", "txpower", "channel", "enabled", "rxprob", "txprob", "int_id"):
        setattr(nd, f, getattr(g["nodes"], f).copy())
    return nd


@pytest.mark.parametrize("name", SCENARIOS)
def test_oracle_reproduces_golden(O, name):
    g = load(name)
    nd = _nodes(O, g)
    kw = dict(g["params"])
    if g["matrix"] is not None:
        kw["n2n_matrix"] = g["matrix"]
    mdl = O.model(KINDS[g["kind"]], **kw)
    state = O.lib().orc_jrandom_seed(g["seed"]) if g["seed"] is not None else 0
    sinr_mode = g["kind"] == "logdist" and g["params"].get("ld_flags", 0) & 1
    onair = np.zeros(0, dtype=O.PACKET_DTYPE)
    for t in g["ticks"]:
        onair = onair[onair["start_us"] + onair["air_us"] > t["begin"]] if sinr_mode else onair[:0]
        active = np.concatenate([onair, t["packets"].astype(O.PACKET_DTYPE)])
        r = O.tick(mdl, nd, active, first_new=len(onair), rng_state=state)
        state = r.rng_state
        for f in ("pkt", "dst", "verdict", "rssi", "sinr"):
            np.testing.assert_array_equal(getattr(r, f), t[f], err_msg="%s %s" % (name, f))
        np.testing.assert_array_equal(r.pkt_interference, t["interference"])
        onair = active
    assert state == g["final_rng_state"]


def test_detmath_golden_bits(O):
    z = np.load(os.path.join(GOLDEN, "detmath.npz"))
    L = O.lib()
    for fn, x, y in ((L.orc_det_log2, "log2_x", "log2_y"), (L.orc_det_exp2, "exp2_x", "exp2_y"),
                     (L.orc_det_normal, "normal_u", "normal_g"), (L.orc_fixed_roundtrip, "fixed_x", "fixed_y")):
        got = np.array([fn(float(v)) for v in z[x]])
        np.testing.assert_array_equal(got.view(np.uint64), z[y].view(np.uint64), err_msg=x)
    h = np.array([L.orc_shadow_hash(int(z["hash_seed"]), int(a), int(b)) for a, b in z["hash_pairs"]], dtype=np.uint64)
    np.testing.assert_array_equal(h, z["hash_h"])


def test_detmath_against_the_rational_arithmetic_fixture(O, rsa):
    """tests/golden/detmath_mp.npz comes from a third, structurally different evaluation of the extension spec
    (tests/golden/make_detmath_mp.py: exact rational arithmetic, one rounding per operation, constants derived with
    mpmath, accuracy checked against the true functions).  Both C texts must agree with it bit for bit: the oracle,
    and the kernels' own header as the host compiler builds it (rm_det_math / rm_link_hash, no device needed)."""
    import ctypes as C
    from radio_sim_amd import _lib
    z = np.load(os.path.join(GOLDEN, "detmath_mp.npz"))
    L, P = O.lib(), _lib.lib()
    cases = (("log2_x", "log2_y", L.orc_det_log2, 0, 1.0), ("log2_x", "log10_y", L.orc_det_log10, 2, 1.0),
             ("exp2_x", "exp2_y", L.orc_det_exp2, 1, 1.0), ("exp2_x", "pow10_y", L.orc_det_pow10, 3, 0.1),
             ("normal_u", "normal_g", L.orc_det_normal, 4, 1.0), ("fixed_x", "fixed_y", L.orc_fixed_roundtrip, 5, 1.0))
    for xk, yk, ofn, pfn, scale in cases:
        xs = z[xk] * scale if scale == 1.0 else z[xk] / 10.0
        want = z[yk].view(np.uint64)
        got_o = np.array([ofn(float(v)) for v in xs]).view(np.uint64)
        got_p = np.array([P.rm_det_math(pfn, float(v)) for v in xs]).view(np.uint64)
        np.testing.assert_array_equal(got_o, want, err_msg="oracle " + yk)
        np.testing.assert_array_equal(got_p, want, err_msg="engine header " + yk)
    seed = int(z["hash_seed"])
    u = C.c_double(0.0)
    for (a, b), h, uu in zip(z["hash_pairs"], z["hash_h"], z["hash_u"]):
        assert L.orc_shadow_hash(seed, int(a), int(b)) == int(h)
        assert P.rm_link_hash(seed, int(a), int(b), C.byref(u)) == int(h) and u.value == float(uu)
        assert P.rm_link_hash(seed, int(b), int(a), None) == int(h)       # symmetric per link


def test_detmath_accuracy(O):
    """E-math of the extension spec against libm / scipy: a few ulp, not bit-exactness."""
    from scipy.stats import norm
    L = O.lib()
    rng = np.random.default_rng(1)
    x = np.exp(rng.uniform(-80, 80, 20000))
    got = np.array([L.orc_det_log2(v) for v in x])
    assert np.max(np.abs(got - np.log2(x)) / np.maximum(np.abs(np.log2(x)), 1.0)) < 4e-16
    y = rng.uniform(-300, 300, 20000)
    got = np.array([L.orc_det_exp2(v) for v in y])
    assert np.max(np.abs(got - np.exp2(y)) / np.exp2(y)) < 5e-16
    assert L.orc_det_log2(1.0) == 0.0 and L.orc_det_log2(1024.0) == 10.0 and L.orc_det_exp2(10.0) == 1024.0
    assert L.orc_det_exp2(-2000.0) == 0.0 and L.orc_det_exp2(2000.0) == np.inf
    for v in (1e-9, 0.37, 1.0, 12.5, 123456.789):
        assert abs(L.orc_det_log10(v) - np.log10(v)) <= 4e-16 * max(1.0, abs(np.log10(v)))
        assert abs(L.orc_det_pow10(np.log10(v)) - v) <= 1e-14 * v
    u = np.concatenate([rng.uniform(0, 1, 20000), [1e-12, 1 - 1e-12]])
    g = np.array([L.orc_det_normal(v) for v in u])
    assert np.max(np.abs(g - norm.ppf(u)) / np.maximum(1.0, np.abs(g))) < 2e-9      # Acklam: 1.15e-9
    assert np.all(np.diff(np.array([L.orc_det_normal(v) for v in np.linspace(1e-6, 1 - 1e-6, 5000)])) > 0)


def test_shadowing_deviate_statistics(O):
    """symmetric, seeded, clipped, ~N(0,1)."""
    m = O.model(O.MODEL_LOGDIST, ld_sigma_db=4.0, ld_seed=7, ld_clip=3.0)
    import ctypes as C
    L = O.lib()
    a = np.arange(0, 300)
    g = np.array([[L.orc_shadow_gauss(C.byref(m), int(i), int(j)) for j in a[:100]] for i in a])
    assert L.orc_shadow_gauss(C.byref(m), 5, 9) == L.orc_shadow_gauss(C.byref(m), 9, 5)
    assert np.abs(g).max() <= 3.0
    off = g[100:, :]     # pairs with i != j
    assert abs(off.mean()) < 0.03 and abs(off.std() - 0.9866) < 0.03     # std of N(0,1) clipped at 3
    assert abs(np.corrcoef(off[:, 0], off[:, 1])[0, 1]) < 0.2
    m2 = O.model(O.MODEL_LOGDIST, ld_sigma_db=4.0, ld_seed=8, ld_clip=3.0)
    assert L.orc_shadow_gauss(C.byref(m2), 5, 9) != L.orc_shadow_gauss(C.byref(m), 5, 9)


def test_fixed_point_sum_is_order_independent(O):
    L = O.lib()
    assert L.orc_fixed_roundtrip(1.0) == 1.0 and L.orc_fixed_roundtrip(2.0 ** -80) == 2.0 ** -80
    assert L.orc_fixed_roundtrip(2.0 ** -81) == 0.0
    v = 1e-10 * (1 + 2.0 ** -30)
    assert abs(L.orc_fixed_roundtrip(v) - v) <= 2.0 ** -80


def test_logdist_known_values(O):
    """hand-checkable: rssi = txpower - pl0 - 10 n log10(d/d0), d0 clamp, sensitivity edge."""
    nd = O.NodeTable(4)
    nd.x[:] = [0.0, 10.0, 100.0, 0.5]
    m = O.model(O.MODEL_LOGDIST)                      # pl0 40, n 3, d0 1, sens -95
    r = O.tick(m, nd, nd.packet(0))
    assert list(r.dst) == [1, 3]                      # 100 m: -100 dBm < -95 -> unheard
    assert abs(r.rssi[0] - (-70.0)) < 1e-12           # 10 m: 0 - 40 - 30
    assert r.rssi[1] == -40.0                         # 0.5 m < d0: clamped to d0 -> log10(1) = 0 exactly
    r = O.tick(m, nd, nd.packet(0, txpower=5.0))
    assert list(r.dst) == [1, 2, 3] and abs(r.rssi[1] - (-95.0)) < 1e-12
