"""Known-answer tests K1..K10 of SURVEY.md section 8c: they pin the CPU oracle to the reference's
Java source (the reference itself ships no tests or golden vectors for this path).
Reference paths: /root/reference/radio-medium/java/se/sics/emul8/radiomedium/."""
import ctypes as C

import numpy as np
import pytest


def test_k1_java_lcg(O):
    # java.util.Random: new Random(42).nextInt() == -1170105035 (well-known answer)
    assert O.JavaRandom(42).next_int() == -1170105035
    r = O.JavaRandom(42)
    assert [r.next_double() for _ in range(5)] == [0.7275636800328681, 0.6832234717598454, 0.30871945533265976,
                                                   0.27707849007413665, 0.6655489517945736]
    r = O.JavaRandom(0)
    assert [r.next_double() for _ in range(3)] == [0.730967787376657, 0.24053641567148587, 0.6374174253501083]


def _three(O, pts):
    nd = O.NodeTable(len(pts))
    for i, p in enumerate(pts):
        nd.x[i], nd.y[i], nd.z[i] = p
    return nd


def test_k2_udgm_boundary(O):
    # UDGMRadioMedium.java:75-78: ratio > 1 -> unheard, so d == range is IN range
    nd = _three(O, [(0, 0, 0), (30, 40, 0), (30, 40, 0.001)])
    res = O.tick(O.model(O.MODEL_UDGM), nd, nd.packet(0))
    assert list(res.dst) == [1] and list(res.verdict) == [O.DELIVERED]


def test_k3_constant_loss_boundary(O):
    # UDGMConstantLossRadioMedium.java:30: strict distance < range(100)
    nd = _three(O, [(0, 0, 0), (60, 80, 0), (59.999, 80, 0)])
    assert O.lib().orc_distance(0, 0, 0, 60, 80, 0) == 100.0
    res = O.tick(O.model(O.MODEL_UDGM_CONST), nd, nd.packet(0))
    assert list(res.dst) == [2] and list(res.verdict) == [O.DELIVERED]


def test_k4_udgm_probability(O):
    nd = _three(O, [(0, 0, 0), (30, 40, 0), (25, 0, 0)])
    nd.rxprob[2] = 0.9
    m = O.model(O.MODEL_UDGM, udgm_success_ratio_rx=0.5)
    ns = nd.as_struct()
    pk = nd.packet(0)
    p = O.Packet.from_buffer_copy(pk.tobytes())
    L = O.lib()
    assert L.orc_udgm_rx_probability(C.byref(m), C.byref(ns), C.byref(p), 1) == 0.5
    assert L.orc_udgm_rx_probability(C.byref(m), C.byref(ns), C.byref(p), 2) == (1 - 0.25 * 0.5) * 0.9 == 0.7875


def test_k5_filters(O):
    nd = _three(O, [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0)])
    nd.channel[1] = 25      # channel mismatch (UDGMRadioMedium.java:102)
    nd.enabled[2] = 0       # radio-state disabled
    for kind in (O.MODEL_NULL, O.MODEL_UDGM, O.MODEL_UDGM_CONST):
        res = O.tick(O.model(kind), nd, nd.packet(0))
        assert list(res.dst) == [3], kind   # source skipped, 1 and 2 filtered


def test_k6_draw_accounting_short_circuit(O):
    # seed 42, successRatioRx = 0.5: txSuccess = 0.5 -> one draw 0.7275.. > 0.5 -> interference;
    # then every in-range receiver is interfered and NO further draw is consumed (:106)
    nd = _three(O, [(0, 0, 0), (10, 0, 0), (20, 0, 0), (30, 0, 0), (500, 0, 0)])
    m = O.model(O.MODEL_UDGM, udgm_success_ratio_rx=0.5)
    res = O.tick(m, nd, nd.packet(0), rng_state=O.lib().orc_jrandom_seed(42))
    assert list(res.dst) == [1, 2, 3] and list(res.verdict) == [O.INTERFERED] * 3
    assert list(res.pkt_interference) == [1] and list(res.pkt_draws) == [1]
    chk = O.JavaRandom(42)
    chk.next_double()
    assert res.rng_state == chk.state.value
    # txSuccess >= 1: no Tx draw; receivers with p < 1 draw in node order
    nd.rxprob[:] = [1.0, 0.5, 1.0, 0.5, 1.0]
    m = O.model(O.MODEL_UDGM)
    res = O.tick(m, nd, nd.packet(0), rng_state=O.lib().orc_jrandom_seed(42))
    assert list(res.pkt_draws) == [2]
    # draws 0.7275 > 0.5 -> interfered ; 0.6832 > 0.5 -> interfered
    assert list(res.dst) == [1, 2, 3] and list(res.verdict) == [O.INTERFERED, O.DELIVERED, O.INTERFERED]


def test_k7_default_config_consumes_no_draws(O):
    nd = _three(O, [(0, 0, 0), (10, 0, 0), (20, 0, 0)])
    s0 = O.lib().orc_jrandom_seed(7)
    res = O.tick(O.model(O.MODEL_UDGM), nd, nd.packets([0, 1, 2]), rng_state=s0)
    assert res.rng_state == s0 and list(res.pkt_draws) == [0, 0, 0]


def test_k8_tx_ratio_unused(O):
    # getTxSuccessProbability uses successRatioRx (UDGMRadioMedium.java:63-65); successRatioTx is dead
    nd = _three(O, [(0, 0, 0), (10, 0, 0)])
    a = O.tick(O.model(O.MODEL_UDGM, udgm_success_ratio_tx=0.0), nd, nd.packet(0), rng_state=1)
    b = O.tick(O.model(O.MODEL_UDGM), nd, nd.packet(0), rng_state=1)
    assert list(a.verdict) == list(b.verdict) == [O.DELIVERED] and a.rng_state == b.rng_state == 1


def test_k9_air_time_and_event_times(O):
    L = O.lib()
    assert L.orc_air_time_us(len("0102030405")) == 320      # RadioPacket.java:72: 32 us per hex char
    t0, t1 = C.c_int64(), C.c_int64()
    L.orc_event_times(1000, 320, 5000, C.byref(t0), C.byref(t1))   # Simulator.java:323-326
    assert (t0.value, t1.value) == (5000, 5320)
    L.orc_event_times(7000, 320, 5000, C.byref(t0), C.byref(t1))
    assert (t0.value, t1.value) == (7000, 7320)


def test_k10_operation_order(O):
    # the verdict uses sqrt THEN square; sqrt(s)^2 != s for roughly half of random pairs
    rng = np.random.default_rng(10)
    a = rng.uniform(0, 100, (20000, 2))
    diff = 0
    for x, y in a:
        d = O.lib().orc_distance(0, 0, 0, x, y, 0)
        diff += (d * d) != (x * x + y * y)
    assert 0.3 < diff / len(a) < 0.7


def test_n2n_semantics(O):
    # N2NRadioMedium.java:28-37: 1-based ids, out-of-range / non-numeric ids -> 0
    nd = _three(O, [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0)])
    nd.int_id[:] = [1, 2, -1, 5]
    m = np.array([[0.0, 1.0, 1.0], [1.0, 0.0, 1.0], [1.0, 1.0, 0.0]])
    res = O.tick(O.model(O.MODEL_N2N, n2n_matrix=m), nd, nd.packet(0), rng_state=3)
    assert list(res.dst) == [1] and list(res.verdict) == [O.DELIVERED]
    res = O.tick(O.model(O.MODEL_N2N, n2n_matrix=m), nd, nd.packet(2), rng_state=3)   # source id -1
    assert res.count == 0


def test_null_medium_everyone_same_channel(O):
    nd = _three(O, [(0, 0, 0), (1e6, 0, 0), (2, 0, 0)])
    res = O.tick(O.model(O.MODEL_NULL), nd, nd.packet(1, txpower=-3.5))
    assert list(res.dst) == [0, 2] and list(res.rssi) == [-3.5, -3.5]      # rssi = packet txpower (:57)


def test_threaded_pass_equals_the_serial_pass_without_draws(O):
    """orc_tick_mt spreads the packets of a tick over threads; that is the reference's result exactly when no
    java.util.Random draw happens (the shared generator is all that chains the packets): equal to orc_tick bit for bit
    with the SINR extension and frames on the air, refused when a link would draw."""
    n = 6000
    rng = np.random.default_rng(1)
    nd = O.NodeTable(n)
    side = 50 * np.sqrt(np.pi * n / 20)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.channel[:] = 11 + rng.integers(0, 3, n)
    nd.enabled[rng.random(n) < 0.05] = 0
    nd.rxprob[rng.random(n) < 0.05] = 0.0        # 0 and 1 draw nothing
    nd.txprob[rng.random(n) < 0.05] = 0.0
    old = nd.packets(np.sort(rng.choice(n, 150, replace=False)), -4000, 8128)
    new = nd.packets(np.sort(rng.choice(n, 200, replace=False)), 0, 8128)
    new["start_us"] = rng.integers(0, 1000, len(new))
    active = np.concatenate([old, new])
    for kind, kw in ((O.MODEL_LOGDIST, dict(ld_sigma_db=4.0, ld_seed=3, ld_flags=1)), (O.MODEL_LOGDIST, dict(ld_sigma_db=0.0)),
                     (O.MODEL_UDGM, {}), (O.MODEL_UDGM_CONST, {}), (O.MODEL_NULL, {})):
        mdl = O.model(kind, **kw)
        a = O.tick(mdl, nd, active, first_new=len(old))
        for threads in (1, 3, 0):
            b = O.tick_mt(mdl, nd, active, first_new=len(old), threads=threads)
            assert a.count == b.count and a.count > 0
            for f in ("pkt", "dst", "verdict", "rssi", "sinr", "pkt_interference"):
                np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)
        assert a.pkt_draws.sum() == 0
    nd.rxprob[:] = np.where(rng.random(n) < 0.5, 0.5, 1.0)
    with pytest.raises(ValueError):
        O.tick_mt(O.model(O.MODEL_UDGM), nd, new)
