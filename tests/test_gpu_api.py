"""C-ABI behaviour beyond the evaluation itself: node updates (the reference has no change hook, so
the shim re-uploads or updates nodes), error codes, call-sequence rules, capacity handling."""
import ctypes as C

import numpy as np
import pytest

from util import configure_engine, oracle_model, to_tx_records, random_nodes, assert_same

pytestmark = pytest.mark.gpu


def test_node_update_moves_a_receiver_and_resorts(engine, rsa, O):
    n = 2000
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=1)
    configure_engine(engine, nd, "udgm", {})
    mdl = oracle_model(O, "udgm", {})
    srcs = np.arange(0, n, 25)
    rng = np.random.default_rng(3)
    for step in range(4):
        # node-config-set: new position / channel / radio-state / rx-loss for a few nodes
        for i in rng.choice(n, 20, replace=False):
            nd.x[i], nd.y[i] = rng.uniform(0, 400, 2)
            nd.channel[i] = 26 if rng.random() < 0.8 else 25
            nd.enabled[i] = 0 if rng.random() < 0.2 else 1
            nd.txpower[i] = -float(step)
            engine.update_node(int(i), nd.x[i], nd.y[i], nd.z[i], nd.txpower[i], int(nd.channel[i]), int(nd.enabled[i]),
                               nd.rxprob[i], nd.txprob[i])
        if step == 2:          # far outside the old frame: the fp32 frame is recomputed
            nd.x[5] = 1.0e6
            engine.update_node(5, nd.x[5], nd.y[5], nd.z[5], nd.txpower[5], int(nd.channel[5]), int(nd.enabled[5]),
                               nd.rxprob[5], nd.txprob[5])
        pk = nd.packets(srcs)
        gpu = engine.tick(to_tx_records(rsa, pk))
        cpu = O.tick(mdl, nd, pk)
        assert_same(gpu, cpu, "after update %d" % step)


@pytest.mark.parametrize("kind,params", [("udgm", {}), ("logdist", dict(ld_sigma_db=4.0, ld_seed=9))])
def test_moved_nodes_are_written_in_place_and_the_table_is_sorted_again_when_it_pays(engine, rsa, O, kind, params):
    """Mobility: node-config-set with a new position per node (rm_node_update) or the shim's dirty list in
    one call (rm_nodes_move).  Small moves keep the receiver table's order; results never depend on it."""
    n = 20000
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd = random_nodes(O, n, side, seed=11)
    configure_engine(engine, nd, kind, params)
    mdl = oracle_model(O, kind, params)
    rng = np.random.default_rng(5)
    srcs = np.sort(rng.choice(n, 200, replace=False))
    engine.tick(to_tx_records(rsa, nd.packets(srcs)))
    builds0 = engine.receiver_table_builds()
    assert builds0 == 1
    for step in range(4):
        # a random walk of 300 nodes (a few metres), flushed in one call; z == None keeps the plane z = 0
        who = rng.choice(n, 300, replace=False).astype(np.int32)
        nd.x[who] += rng.normal(0, 3.0, who.size)
        nd.y[who] += rng.normal(0, 3.0, who.size)
        engine.move_nodes(who, nd.x[who], nd.y[who])
        # and some single updates, transmitters among them
        for i in list(rng.choice(n, 5, replace=False)) + [int(srcs[step])]:
            nd.x[i] += rng.normal(0, 3.0)
            nd.channel[i] = 26 if rng.random() < 0.7 else 25
            engine.update_node(int(i), nd.x[i], nd.y[i], nd.z[i], nd.txpower[i], int(nd.channel[i]), int(nd.enabled[i]),
                               nd.rxprob[i], nd.txprob[i])
        pk = nd.packets(srcs)
        assert_same(engine.tick(to_tx_records(rsa, pk)), O.tick(mdl, nd, pk), "random walk %d" % step)
    assert engine.receiver_table_builds() == builds0, "small moves must not sort the table again"
    # teleports across the field: correct at once, and after enough of them the table is sorted again
    for step in range(3):
        who = rng.choice(n, 40, replace=False).astype(np.int32)
        nd.x[who] = rng.uniform(0, side, who.size)
        nd.y[who] = rng.uniform(0, side, who.size)
        nd.z[who] = rng.uniform(0, 5.0, who.size)
        engine.move_nodes(who, nd.x[who], nd.y[who], nd.z[who])
        pk = nd.packets(srcs)
        assert_same(engine.tick(to_tx_records(rsa, pk)), O.tick(mdl, nd, pk), "teleport %d" % step)
    assert engine.receiver_table_builds() > builds0
    # the per-packet call sees the same table
    i = int(srcs[7])
    one = engine.transmit(i, 0, 254)
    ref = O.tick(mdl, nd, nd.packets(np.array([i])))
    np.testing.assert_array_equal(one.dst, ref.dst)
    np.testing.assert_array_equal(one.verdict, ref.verdict)


def test_moved_nodes_with_a_receiver_partition_and_probabilities(engine, rsa, O):
    n = 6000
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=12)
    configure_engine(engine, nd, "udgm", {})
    mdl = oracle_model(O, "udgm", {})
    engine.set_partition(2000, 3000)
    rng = np.random.default_rng(6)
    srcs = np.sort(rng.choice(n, 120, replace=False))
    for step in range(3):
        who = rng.choice(n, 200, replace=False).astype(np.int32)   # receivers of this partition and of others, sources
        nd.x[who] += rng.normal(0, 10.0, who.size)
        nd.y[who] += rng.normal(0, 10.0, who.size)
        engine.move_nodes(who, nd.x[who], nd.y[who])
        pk = nd.packets(srcs)
        gpu = engine.tick(to_tx_records(rsa, pk))
        cpu = O.tick(mdl, nd, pk)
        keep = (cpu.dst >= 2000) & (cpu.dst < 5000)
        np.testing.assert_array_equal(gpu.pkt, cpu.pkt[keep])
        np.testing.assert_array_equal(gpu.dst, cpu.dst[keep])
        np.testing.assert_array_equal(gpu.verdict, cpu.verdict[keep])
    # rx-loss on one node makes the medium draw; taking it away again stops the draws (the cached answer follows)
    engine.set_partition(0, n)
    engine.seed(42)
    i = int(srcs[3]) + 1
    nd.rxprob[i] = 0.5
    engine.update_node(i, nd.x[i], nd.y[i], nd.z[i], nd.txpower[i], int(nd.channel[i]), int(nd.enabled[i]), nd.rxprob[i], nd.txprob[i])
    pk = nd.packets(srcs)
    cpu = O.tick(mdl, nd, pk, rng_state=O.lib().orc_jrandom_seed(42))
    assert_same(engine.tick(to_tx_records(rsa, pk)), cpu, "after rx-loss")
    assert engine.rng_state == cpu.rng_state
    nd.rxprob[i] = 1.0
    engine.update_node(i, nd.x[i], nd.y[i], nd.z[i], nd.txpower[i], int(nd.channel[i]), int(nd.enabled[i]), nd.rxprob[i], nd.txprob[i])
    before = engine.rng_state
    assert_same(engine.tick(to_tx_records(rsa, pk)), O.tick(mdl, nd, pk), "rx-loss removed")
    assert engine.rng_state == before


def test_nodes_move_rejects_bad_input(engine, rsa, O):
    from radio_sim_amd import _lib
    nd = random_nodes(O, 100, 300.0, seed=2)
    configure_engine(engine, nd, "udgm", {})
    with pytest.raises(_lib.RadioMediumError):
        engine.move_nodes([100], [0.0], [0.0])
    with pytest.raises(_lib.RadioMediumError):
        engine.move_nodes([3], [float("nan")], [0.0])
    engine.move_nodes(np.zeros(0, np.int32), np.zeros(0), np.zeros(0))   # an empty dirty list


@pytest.mark.parametrize("kind,params,n,t", [
    ("udgm", {}, 3000, 10),
    ("udgm", {"udgm_success_ratio_rx": 0.8}, 20000, 300),          # draws
    ("logdist", dict(ld_sigma_db=4.0, ld_seed=3), 20000, 5000),     # more links and packets than the first block holds
    ("logdist", dict(ld_sigma_db=4.0, ld_seed=3, ld_flags=1), 5000, 200),   # SINR
    ("null", {}, 300, 3),                                           # unsorted table
])
def test_flush_into_the_pinned_block_equals_the_oracle(engine, rsa, O, kind, params, n, t):
    """rm_tick_flush_view: the result is read in place from the engine's host-mapped block;
    rm_tick_flush copies from the same block."""
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=n + t)
    configure_engine(engine, nd, kind, params)
    mdl = oracle_model(O, kind, params)
    rng = np.random.default_rng(t)
    state = O.lib().orc_jrandom_seed(5)
    engine.seed(5)
    for step in range(3):
        srcs = np.sort(rng.choice(n, t, replace=False))
        pk = nd.packets(srcs, 1000 * step, 320)
        cpu = O.tick(mdl, nd, pk, rng_state=state)
        state = cpu.rng_state
        engine.tick_begin(1000 * step, 1000 * step + 1000)
        engine.enqueue_records(to_tx_records(rsa, pk))
        gpu = engine.tick_flush_view() if step != 1 else engine.tick_flush()
        assert_same(gpu, cpu, "step %d" % step)
        np.testing.assert_array_equal(gpu.pkt_interference, cpu.pkt_interference)
        np.testing.assert_array_equal(np.diff(gpu.pkt_offset.astype(np.int64)), np.bincount(cpu.pkt, minlength=t))
        assert engine.rng_state == state
    # an empty tick
    engine.tick_begin(9000, 10000)
    empty = engine.tick_flush_view()
    assert empty.count == 0 and len(empty.pkt_offset) == 1 and empty.pkt_offset[0] == 0


def test_flush_view_reports_capacity_like_the_copying_flush(engine, rsa, O):
    from radio_sim_amd import _lib
    n = 4000
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=8)
    configure_engine(engine, nd, "udgm", {})
    engine.set_link_capacity(256)
    pk = nd.packets(np.arange(0, n, 4))
    engine.tick_begin(0, 1000)
    engine.enqueue_records(to_tx_records(rsa, pk))
    with pytest.raises(_lib.RadioMediumError) as e:
        engine.tick_flush_view()
    assert e.value.code == _lib.RM_ERR_CAPACITY


def test_error_codes_and_call_sequence(engine, rsa, O):
    from radio_sim_amd import _lib
    L = _lib.lib()
    h = engine._h
    nd = random_nodes(O, 100, 100.0, seed=2)
    engine.upload_table(nd)
    assert L.rm_node_count(h) == 100
    p = _lib.ModelParams()
    L.rm_model_defaults(C.byref(p), 99)
    assert L.rm_set_model(h, C.byref(p)) == _lib.RM_ERR_INVALID and b"unknown model" in L.rm_last_error()
    L.rm_model_defaults(C.byref(p), 4)
    p.ld_d0 = 0.0
    assert L.rm_set_model(h, C.byref(p)) == _lib.RM_ERR_INVALID
    assert L.rm_enqueue_tx(h, 1, 0, 0, None, None) == _lib.RM_ERR_STATE            # outside a tick
    assert L.rm_tick_flush(h, None, None, None, None, None, 0, None, None, None) == _lib.RM_ERR_STATE
    assert L.rm_tick_begin(h, 0, 1000) == 0
    assert L.rm_enqueue_tx(h, 100, 0, 0, None, None) == _lib.RM_ERR_INVALID        # "could not find source node"
    assert L.rm_enqueue_tx(h, 3, 0, -5, None, None) == _lib.RM_ERR_INVALID
    assert L.rm_set_partition(h, 50, 51) == _lib.RM_ERR_INVALID
    assert L.rm_node_update(h, 100, 0.0, 0.0, 0.0, 0.0, 26, 1, 1.0, 1.0) == _lib.RM_ERR_INVALID
    assert L.rm_node_update(h, 1, float("nan"), 0.0, 0.0, 0.0, 26, 1, 1.0, 1.0) == _lib.RM_ERR_INVALID
    x = np.array([0.0, np.inf])
    assert L.rm_nodes_upload(h, 2, x.ctypes.data, x.ctypes.data, None, None, None, None, None, None, None) == _lib.RM_ERR_INVALID
    assert L.rm_create(7, C.byref(C.c_void_p())) == _lib.RM_ERR_INVALID            # device ordinal out of range
    assert L.rm_result_count(h, None, None) in (0, _lib.RM_ERR_STATE)


def test_caller_buffer_smaller_than_the_links(engine, rsa, O):
    nd = random_nodes(O, 500, 150.0, seed=4)
    configure_engine(engine, nd, "udgm", {})
    recs = to_tx_records(rsa, nd.packets([1, 2, 3]))
    full = engine.tick(recs)
    assert full.count > 50
    engine.tick_begin(0, 0)
    engine.enqueue_records(recs)
    with pytest.raises(rsa.RadioMediumError) as e:
        engine.tick_flush(cap=10)                   # count is still reported through the error path
    assert e.value.code == -4
    again = engine.tick(recs)                       # the context stays usable
    assert again.count == full.count


def test_defaults_of_optional_arrays(engine, rsa, O):
    """NULL arrays in rm_nodes_upload mean the reference's field defaults (Transciever.java:11-18)."""
    x = np.array([0.0, 10.0, 200.0])
    y = np.zeros(3)
    engine.upload_nodes(x, y)
    engine.set_model(rsa.MODEL_UDGM)
    r = engine.transmit(0, start_us=5, hex_length=10)
    assert list(r.dst) == [1] and list(r.verdict) == [rsa.DELIVERED] and list(r.rssi) == [0.0]
    engine.set_model(rsa.MODEL_N2N)
    engine.set_n2n_matrix(np.array([[0, 1, 1], [1, 0, 1], [1, 1, 0]], dtype=float))   # int_id default = index + 1
    r = engine.transmit(2)
    assert list(r.dst) == [0, 1]


def test_transmit_fast_path_and_its_fallbacks(engine, rsa, O):
    """rm_transmit returns through one host-mapped block of 2048 links; more links than that, a
    caller buffer smaller than the links, and per-packet overrides all keep the reference's answers."""
    n = 3000
    nd = random_nodes(O, n, 300.0, seed=8)
    nd.rxprob[::5] = 0.5
    # Null medium: every other node hears the packet (NullRadioMedium.java:47-77) -> 2999 links > 2048
    configure_engine(engine, nd, "null", {})
    got = engine.transmit(7, start_us=5, hex_length=20)
    want = O.tick(oracle_model(O, "null", {}), nd, nd.packets([7], 5, 640))
    assert got.count == want.count == n - 1
    np.testing.assert_array_equal(got.dst, want.dst)
    np.testing.assert_array_equal(got.verdict, want.verdict)
    np.testing.assert_array_equal(got.rssi, want.rssi)
    # UDGM with draws: the generator advances exactly as in the oracle, call after call
    params = dict(udgm_success_ratio_rx=0.9)
    configure_engine(engine, nd, "udgm", params)
    engine.seed(3)
    state = O.lib().orc_jrandom_seed(3)
    mdl = oracle_model(O, "udgm", params)
    for src, power, ch in ((11, None, None), (12, -7.5, None), (13, None, 26), (2999, -1.0, 26)):
        got = engine.transmit(src, start_us=100, hex_length=254, txpower=power, channel=ch)
        pk = nd.packets([src], 100, 8128)
        if power is not None:
            pk["txpower"] = power
        if ch is not None:
            pk["channel"] = ch
        want = O.tick(mdl, nd, pk, rng_state=state)
        state = want.rng_state
        assert got.count == want.count > 0
        np.testing.assert_array_equal(got.dst, want.dst)
        np.testing.assert_array_equal(got.verdict, want.verdict)
        np.testing.assert_array_equal(got.rssi, want.rssi)
        assert bool(got.pkt_interference[0]) == bool(want.pkt_interference[0])
        assert engine.rng_state == state
    with pytest.raises(rsa.RadioMediumError) as e:
        engine.transmit(11, hex_length=254, cap=3)
    assert e.value.code == -4


@pytest.mark.parametrize("kind,params", [("udgm", {}), ("udgm", dict(udgm_transmission_range=600.0, udgm_success_ratio_rx=0.6)),
                                         ("udgm_const", {}), ("logdist", dict(ld_sigma_db=4.0, ld_seed=21)),
                                         ("logdist", dict(ld_sigma_db=0.0, ld_exponent=0.0))])
def test_one_launch_transmit_against_the_oracle(engine, rsa, O, kind, params):
    """k_transmit_one (and its hand-over to the general path: > 2048 links with the 600 m range,
    no geometric bound with a path-loss exponent of 0) over many packets, with disabled radios, other
    channels, 3-D positions, fractional probabilities and a receiver partition."""
    n = 6000
    rng = np.random.default_rng(5)
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=6, z_span=30.0)
    nd.enabled[rng.choice(n, 200, replace=False)] = 0
    nd.channel[rng.choice(n, 600, replace=False)] = 25
    nd.rxprob[rng.choice(n, 1500, replace=False)] = 0.7
    nd.txprob[rng.choice(n, 300, replace=False)] = 0.4
    nd.txprob[rng.choice(n, 50, replace=False)] = 0.0
    configure_engine(engine, nd, kind, params)
    mdl = oracle_model(O, kind, params)
    engine.seed(1234)
    state = O.lib().orc_jrandom_seed(1234)
    for src in rng.choice(n, 60, replace=False):
        got = engine.transmit(int(src), start_us=7, hex_length=100)
        want = O.tick(mdl, nd, nd.packets([int(src)], 7, 3200), rng_state=state)
        state = want.rng_state
        assert got.count == want.count, (kind, src)
        np.testing.assert_array_equal(got.dst, want.dst)
        np.testing.assert_array_equal(got.verdict, want.verdict)
        np.testing.assert_array_equal(got.rssi, want.rssi)
        assert bool(got.pkt_interference[0]) == bool(want.pkt_interference[0])
    assert engine.rng_state == state
    # A receiver partition.  With probabilistic links the per-rank draw exchange is needed first, and the
    # per-packet call says so; without them (probabilities 0 or 1 only) the partition hears exactly its
    # share of the oracle's links, through the one-launch kernel.
    engine.set_partition(1000, 3000)
    if kind != "udgm_const":                  # the constant-loss medium never draws (UDGMConstantLossRadioMedium.java:16-36)
        with pytest.raises(rsa.RadioMediumError) as e:
            engine.transmit(5, hex_length=100)
        assert e.value.code == -5
    nd.rxprob[:] = np.where(nd.rxprob < 1.0, 0.0, 1.0)
    nd.txprob[:] = np.where(nd.txprob < 1.0, 0.0, 1.0)
    plain = {k: v for k, v in params.items() if k != "udgm_success_ratio_rx"}
    configure_engine(engine, nd, kind, plain)
    engine.set_partition(1000, 3000)
    mdl = oracle_model(O, kind, plain)
    for src in (5, 1500, 3999, 5999):
        got = engine.transmit(src, hex_length=100)
        want = O.tick(mdl, nd, nd.packets([src], 0, 3200))
        keep = (want.dst >= 1000) & (want.dst < 4000)
        np.testing.assert_array_equal(got.dst, want.dst[keep])
        np.testing.assert_array_equal(got.verdict, want.verdict[keep])
        np.testing.assert_array_equal(got.rssi, want.rssi[keep])
        assert bool(got.pkt_interference[0]) == bool(want.pkt_interference[0])


def test_plain_c_example(tmp_path):
    """examples/udgm_transmit.c: the ABI driven from plain C (K2's boundary case through rm_transmit)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "radio-sim_amd", "csrc")
    exe = os.path.join(str(tmp_path), "udgm_transmit")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "udgm_transmit.c"), "-L" + lib, "-lradiomedium_hip",
                           "-Wl,-rpath," + lib, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "UDGM Radio Medium: 1 heard, Tx ok" in out.stdout and "node 1: delivered, rssi 0.0" in out.stdout
