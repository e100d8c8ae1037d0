"""Host-side logic of the product that needs no GPU: java.util.Random jump-ahead, workload."""
import ctypes as C

import numpy as np


def test_lcg_jump_matches_stepping(rsa, O):
    from radio_sim_amd import _lib
    L = _lib.lib()
    s0 = O.lib().orc_jrandom_seed(42)
    r = O.JavaRandom(42)
    for k in (0, 1, 2, 3, 17, 1000):
        st = C.c_uint64(L.rm_lcg_jump(s0, 2 * k))
        r2 = O.JavaRandom(42)
        for _ in range(k):
            r2.next_double()
        assert st.value == r2.state.value
        assert L.rm_lcg_next_double(C.byref(st)) == r2.next_double()
    # far jump: compose two jumps
    assert L.rm_lcg_jump(L.rm_lcg_jump(s0, 123456789), 987654321) == L.rm_lcg_jump(s0, 123456789 + 987654321)
    assert r.next_double() == 0.7275636800328681


def test_workload_is_deterministic(rsa):
    from radio_sim_amd import workload as W
    a = W.make_nodes(1000, 3)
    b = W.make_nodes(1000, 3)
    assert np.array_equal(a.x, b.x) and np.array_equal(a.y, b.y)
    side = W.side_length(1000)
    assert 0 <= a.x.min() and a.x.max() < side and np.all(a.z == 0)
    # SplitMix64 known answer (seed 0): first output 0xE220A8397B1DCDAF
    assert int(W.splitmix64(0, 1)[0]) == 0xE220A8397B1DCDAF
    s1 = W.choose_sources(1000, 10, 5, 0)
    s2 = W.choose_sources(1000, 10, 5, 0)
    s3 = W.choose_sources(1000, 10, 5, 1)
    assert np.array_equal(s1, s2) and not np.array_equal(s1, s3)
    assert len(set(s1.tolist())) == 10 and np.all(np.diff(s1) > 0)
    full = W.choose_sources(50, 50, 1, 0)
    assert sorted(full.tolist()) == list(range(50))
    nd = W.make_nodes(5000, 4, channels16=True)
    assert nd.channel.min() >= 11 and nd.channel.max() <= 26 and len(set(nd.channel.tolist())) == 16


def test_expected_neighbours(rsa):
    from radio_sim_amd import workload as W
    nd = W.make_nodes(20000, 2)
    # ~20 in-range neighbours at R = 50 m (minus edge effects)
    i = np.arange(0, 20000, 200)
    d2 = (nd.x[i, None] - nd.x[None, :]) ** 2 + (nd.y[i, None] - nd.y[None, :]) ** 2
    k = (d2 <= 2500).sum(1).mean() - 1
    assert 17 < k < 21


def _product_order(ops):
    """pop order of every ('pop', T) of `ops` with the product's order rule (csrc/rm_evorder.hpp through rm_evq_*):
    sort by (time, ladder, -insertion)."""
    from radio_sim_amd import _lib
    L = _lib.lib()
    o = _lib.EvqOrder()
    L.rm_evq_init(C.byref(o))
    pending, out, nid = [], [], 0
    for kind, t in ops:
        if kind == "add":
            pending.append((t, L.rm_evq_add(C.byref(o), t), -nid))
            nid += 1
        else:
            L.rm_evq_drain(C.byref(o), t)
            pending.sort()
            k = 0
            while k < len(pending) and pending[k][0] < t:
                out.append(-pending[k][2])
                k += 1
            pending = pending[k:]
    return out


def test_event_order_rule_equals_the_reference_queue(rsa, O):
    """The sort key the reception stage uses instead of a queue (time, ladder, reverse insertion) against the
    literal restatement of com/botbox/scheduler/EventQueue.java, ties and the moveTop boundary included."""
    kat = [("add", 10), ("add", 50), ("add", 50), ("pop", 20), ("add", 50), ("add", 50), ("add", 30), ("pop", 100)]
    assert _product_order(kat) == list(O.evq_replay(kat)) == [0, 5, 2, 1, 4, 3]
    for seed in range(120):
        rng = np.random.default_rng(1000 + seed)
        now, ops = 0, []
        for _ in range(int(rng.integers(3, 25))):
            for _ in range(int(rng.integers(0, 80))):
                pick = [now, now, now + 320, now + 8128, now + int(rng.integers(0, 1000)), now + 1000, now + 2000]
                ops.append(("add", int(pick[int(rng.integers(0, len(pick)))])))
            now += int(rng.choice([1000, 1000, 1, 10, 5000]))
            ops.append(("pop", now))
        ops.append(("pop", now + 10 ** 9))
        assert _product_order(ops) == list(O.evq_replay(ops)), seed
