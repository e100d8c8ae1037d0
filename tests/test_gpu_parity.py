"""Parity of the HIP engine (through the C ABI) with the CPU oracle: the four reference media.
Bit-exact on heard sets / verdicts / draw consumption; rssi is the packet's transmit power."""
import os

import numpy as np
import pytest

from util import run_both, assert_same, random_nodes, to_tx_records, configure_engine, oracle_model

pytestmark = pytest.mark.gpu


def _line(O, pts):
    nd = O.NodeTable(len(pts))
    for i, p in enumerate(pts):
        nd.x[i], nd.y[i], nd.z[i] = p
    return nd


def test_get_name_and_base_rssi(engine, rsa):
    assert engine.get_name().startswith("Null radio medium")          # Main.java:66-70 default
    engine.set_model(rsa.MODEL_UDGM)
    assert engine.get_name() == "UDGM Radio Medium"
    assert engine.get_base_rssi(0) == -100.0                          # AbstractRadioMedium.java:38
    engine.set_base_rssi(-91.0)
    assert engine.get_base_rssi(3) == -91.0


def test_k2_k3_boundaries(engine, rsa, O):
    nd = _line(O, [(0, 0, 0), (30, 40, 0), (30, 40, 0.001), (60, 80, 0), (59.999, 80, 0)])
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm", {}, nd.packet(0))
    assert_same(gpu, cpu, "K2")
    assert list(gpu.dst) == [1]
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm_const", {}, nd.packet(0))
    assert_same(gpu, cpu, "K3")
    assert list(gpu.dst) == [1, 2, 4]            # d == 100 exactly is out (strict <)


def test_k5_filters_all_models(engine, rsa, O):
    nd = _line(O, [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0)])
    nd.channel[1] = 25
    nd.enabled[2] = 0
    for kind in ("null", "udgm", "udgm_const"):
        gpu, cpu = run_both(O, rsa, engine, nd, kind, {}, nd.packet(0))
        assert_same(gpu, cpu, kind)
        assert list(gpu.dst) == [3]
    # packet channel override ("wireless-channel", SimulatorJSONHandler.java:87-90)
    gpu, cpu = run_both(O, rsa, engine, nd, "null", {}, nd.packet(0, channel=25, txpower=-7.0))
    assert_same(gpu, cpu, "override")
    assert list(gpu.dst) == [1] and list(gpu.rssi) == [-7.0]


def test_k6_k7_draw_accounting(engine, rsa, O):
    nd = _line(O, [(0, 0, 0), (10, 0, 0), (20, 0, 0), (30, 0, 0), (500, 0, 0)])
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm", {"udgm_success_ratio_rx": 0.5}, nd.packet(0), seed=42)
    assert_same(gpu, cpu, "K6a")
    assert list(gpu.verdict) == [rsa.INTERFERED] * 3 and engine.rng_state == cpu.rng_state
    nd.rxprob[:] = [1.0, 0.5, 1.0, 0.5, 1.0]
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm", {}, nd.packet(0), seed=42)
    assert_same(gpu, cpu, "K6b")
    assert list(gpu.verdict) == [rsa.INTERFERED, rsa.DELIVERED, rsa.INTERFERED]
    assert engine.rng_state == cpu.rng_state
    nd.rxprob[:] = 1.0
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm", {}, nd.packets([0, 1, 2]), seed=7)
    assert_same(gpu, cpu, "K7")
    assert engine.rng_state == O.lib().orc_jrandom_seed(7)


@pytest.mark.parametrize("kind", ["null", "udgm", "udgm_const"])
@pytest.mark.parametrize("n,t,z", [(64, 1, 0.0), (1000, 37, 0.0), (3000, 130, 25.0)])
def test_random_layouts_deterministic(engine, rsa, O, kind, n, t, z):
    if kind == "null" and n > 1000:
        pytest.skip("dense output, covered at n = 1000")
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd = random_nodes(O, n, side, seed=n + t, z_span=z)
    rng = np.random.default_rng(5)
    nd.channel[rng.random(n) < 0.1] = 25
    nd.enabled[rng.random(n) < 0.05] = 0
    nd.txpower[:] = rng.uniform(-25, 0, n)
    src = rng.choice(n, t, replace=False)
    gpu, cpu = run_both(O, rsa, engine, nd, kind, {}, nd.packets(src))
    assert cpu.count > 0
    assert_same(gpu, cpu, "%s n=%d" % (kind, n))


@pytest.mark.parametrize("ratio_rx", [1.0, 0.6])
def test_random_layouts_stochastic_udgm(engine, rsa, O, ratio_rx):
    n, t = 2500, 90
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd = random_nodes(O, n, side, seed=99)
    rng = np.random.default_rng(6)
    nd.rxprob[:] = np.where(rng.random(n) < 0.5, 1.0, rng.uniform(0, 1, n))
    nd.rxprob[rng.random(n) < 0.05] = 0.0
    nd.txprob[:] = np.where(rng.random(n) < 0.5, 1.0, rng.uniform(0, 1.2, n))
    nd.txprob[rng.random(n) < 0.05] = 0.0
    src = rng.choice(n, t, replace=False)
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm", {"udgm_success_ratio_rx": ratio_rx}, nd.packets(src), seed=2024)
    assert cpu.pkt_draws.sum() > 50
    assert_same(gpu, cpu, "udgm stochastic")
    assert engine.rng_state == cpu.rng_state
    # a second tick continues the same generator
    src2 = rng.choice(n, t, replace=False)
    mdl = oracle_model(O, "udgm", {"udgm_success_ratio_rx": ratio_rx})
    cpu2 = O.tick(mdl, nd, nd.packets(src2), rng_state=cpu.rng_state)
    gpu2 = engine.tick(to_tx_records(rsa, nd.packets(src2)))
    assert_same(gpu2, cpu2, "udgm stochastic tick 2")
    assert engine.rng_state == cpu2.rng_state


def test_n2n_matrix(engine, rsa, O):
    n = 300
    rng = np.random.default_rng(8)
    nd = random_nodes(O, n, 100.0, seed=3)
    m = np.where(rng.random((n, n)) < 0.1, rng.uniform(0, 1.3, (n, n)), 0.0)
    nd.int_id[:] = np.arange(1, n + 1)
    nd.int_id[5] = -1           # non-numeric id
    nd.int_id[6] = n + 7        # outside the matrix
    nd.rxprob[10:20] = 0.5
    nd.txprob[30:40] = 0.7
    src = rng.choice(n, 60, replace=False)
    gpu, cpu = run_both(O, rsa, engine, nd, "n2n", {}, nd.packets(src), matrix=m, seed=11)
    assert cpu.count > 100
    assert_same(gpu, cpu, "n2n")
    assert engine.rng_state == cpu.rng_state


def test_single_transmit_api(engine, rsa, O):
    nd = random_nodes(O, 500, 300.0, seed=17)
    configure_engine(engine, nd, "udgm", {})
    mdl = oracle_model(O, "udgm", {})
    for src in (0, 77, 499):
        cpu = O.tick(mdl, nd, nd.packet(src, start_us=123, air_us=320))
        res = engine.transmit(src, start_us=123, hex_length=10)
        assert res.count == cpu.count
        np.testing.assert_array_equal(res.dst, cpu.dst)
        np.testing.assert_array_equal(res.verdict, cpu.verdict)
    with pytest.raises(rsa.RadioMediumError):     # "could not find source node"
        engine.transmit(500)


def test_records_with_their_own_tx_probability(engine, rsa, O):
    """A caller's record may carry a txProbability the node table does not have (every node at 1.0, successRatioRx
    1.0: no draw can come from the table).  The reference draws random.nextDouble() > txSuccess for it all the same
    (UDGMRadioMedium.java:87-92): the frame may fail, and the generator moves."""
    nd = random_nodes(O, 900, 400.0, seed=31)
    for kind, params in (("udgm", {}), ("logdist", {})):
        configure_engine(engine, nd, kind, params)
        mdl = oracle_model(O, kind, params)
        pk = nd.packets(np.arange(0, 900, 9), 0, 320)
        pk["txprob"][::2] = 0.5
        pk["txprob"][5] = 0.0
        engine.seed(7)
        cpu = O.tick(mdl, nd, pk, rng_state=O.lib().orc_jrandom_seed(7))
        assert cpu.pkt_draws.sum() >= 40 and 0 < cpu.pkt_interference.sum() < len(pk)
        gpu = engine.tick(to_tx_records(rsa, pk))
        assert_same(gpu, cpu, kind + " records with their own txprob")
        assert engine.rng_state == cpu.rng_state
        # and the next tick, from the table again, takes the draw-free path
        pk2 = nd.packets(np.arange(1, 900, 9), 1000, 320)
        cpu2 = O.tick(mdl, nd, pk2, rng_state=cpu.rng_state)
        assert_same(engine.tick(to_tx_records(rsa, pk2)), cpu2, kind + " next tick")
        assert engine.rng_state == cpu.rng_state == cpu2.rng_state


@pytest.mark.parametrize("lossy", [False, True])
def test_frames_heard_by_everybody(engine, rsa, O, lossy):
    """Nothing to cull: a transmission range that covers the whole layout, so every frame is heard by every node --
    far more links than a frame's LDS segment of the one-launch tick holds (rm_tick.hip: second pass, room from the
    overflow allocator, node order from a bitmap over the node indices).  With and without java.util.Random draws."""
    n = 3000
    nd = random_nodes(O, n, 400.0, seed=41)
    rng = np.random.default_rng(41)
    if lossy:
        nd.rxprob[:] = np.where(rng.random(n) < 0.6, 1.0, rng.uniform(0, 1, n))
    nd.enabled[rng.choice(n, 30, replace=False)] = 0
    params = {"udgm_transmission_range": 5000.0}
    # (RM_FRAME_TICK=0, the developer knob for the three-launch sweep: its candidate shards follow the receiver tiles, and
    # a table of 12 tiles with everything heard needs far more room per shard)
    engine.set_link_capacity(1 << 20 if os.environ.get("RM_FRAME_TICK") != "0" else 1 << 24)
    pk = nd.packets(rng.choice(n, 40, replace=False), 0, 320)
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm", params, pk, seed=3)
    assert cpu.count > 40 * 2900
    assert_same(gpu, cpu, "everybody in range")
    assert engine.rng_state == cpu.rng_state


def test_frames_heard_by_everybody_large_table(engine, rsa, O):
    """the same on a table too large for the LDS bitmap (170 000 nodes): the frame's links are ordered by counting"""
    n = 170_000
    nd = random_nodes(O, n, 3000.0, seed=42)
    engine.set_link_capacity(1 << 20)
    pk = nd.packets([5, 99_999], 0, 320)
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm", {"udgm_transmission_range": 1e5}, pk)
    assert cpu.count == 2 * (n - 1)
    assert_same(gpu, cpu, "everybody in range, large table")


def test_empty_and_ragged(engine, rsa, O):
    nd = random_nodes(O, 130, 100.0, seed=2)      # not a multiple of 64
    configure_engine(engine, nd, "udgm", {})
    res = engine.tick(np.zeros(0, dtype=rsa.TX_RECORD_DTYPE))
    assert res.count == 0
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm", {}, nd.packets(np.arange(130)))   # every node transmits
    assert_same(gpu, cpu, "ragged")
    nd1 = random_nodes(O, 1, 10.0, seed=1)
    gpu, cpu = run_both(O, rsa, engine, nd1, "null", {}, nd1.packet(0))
    assert gpu.count == cpu.count == 0
    # link capacity smaller than the heard links -> RM_ERR_CAPACITY, count still reported
    configure_engine(engine, nd, "null", {})
    engine.set_link_capacity(100)
    with pytest.raises(rsa.RadioMediumError) as e:
        engine.tick(to_tx_records(rsa, nd.packets([0, 1])))
    assert e.value.code == -4


def test_large_coordinates_use_fp64_filter(engine, rsa, O):
    """Far-from-origin layouts make the fp32 frame too coarse; results must not change."""
    n = 2000
    nd = random_nodes(O, n, 500.0, seed=12)
    nd.x += 3.0e7
    nd.y -= 9.0e7
    rng = np.random.default_rng(1)
    src = rng.choice(n, 64, replace=False)
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm", {}, nd.packets(src))
    assert cpu.count > 500
    assert_same(gpu, cpu, "far origin")
    # one outlier node stretches the frame: still exact
    nd.x[7] = -4.0e9
    gpu, cpu = run_both(O, rsa, engine, nd, "udgm_const", {}, nd.packets(src))
    assert_same(gpu, cpu, "outlier")


def test_boundary_lattice(engine, rsa, O):
    """Integer lattice: many pairs sit exactly on d == range (3-4-5 triples)."""
    g = np.arange(0, 40) * 10.0
    xx, yy = np.meshgrid(g, g)
    nd = O.NodeTable(xx.size)
    nd.x, nd.y = xx.ravel().copy(), yy.ravel().copy()
    src = np.arange(0, nd.n, 23)
    for kind, params in (("udgm", {}), ("udgm_const", {}), ("udgm", {"udgm_transmission_range": 100.0}),
                         ("udgm_const", {"const_range": 50.0})):
        gpu, cpu = run_both(O, rsa, engine, nd, kind, params, nd.packets(src))
        assert_same(gpu, cpu, kind + str(params))


def test_device_resident_paths(engine, rsa, O):
    """rm_pack_tx_device + rm_tick_run_device and the fused rm_tick_run_sources_device give the same
    links as the host-record path / the oracle (sources with -1 padding included)."""
    from util import DeviceArray
    n = 5000
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=44)
    rng = np.random.default_rng(2)
    nd.txpower[:] = rng.uniform(-10, 0, n)
    nd.channel[rng.random(n) < 0.2] = 20
    params = {"ld_sigma_db": 4.0, "ld_seed": 3}
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    srcs = np.sort(rng.choice(n, 100, replace=False)).astype(np.int32)
    padded = np.concatenate([srcs[:50], [-1, -1, -1], srcs[50:]]).astype(np.int32)
    cpu = O.tick(mdl, nd, nd.packets(srcs, 7000, 8128))
    slot_of_valid = np.nonzero(padded >= 0)[0]
    src_dev = DeviceArray(padded)
    rec_dev = DeviceArray(nbytes=len(padded) * 64)
    for fused in (False, True):
        if fused:
            engine.tick_run_sources_device(7000, 8000, src_dev.ptr.value, len(padded), 7000, 8128)
        else:
            engine.pack_tx_device(src_dev.ptr.value, len(padded), 7000, 8128, rec_dev.ptr.value)
            engine.tick_run_device(7000, 8000, rec_dev.ptr.value, len(padded))
        gpu = engine.result_copy(len(padded))
        assert gpu.count == cpu.count > 1000
        np.testing.assert_array_equal(gpu.pkt, slot_of_valid[cpu.pkt])
        np.testing.assert_array_equal(gpu.dst, cpu.dst)
        np.testing.assert_array_equal(gpu.verdict, cpu.verdict)
        np.testing.assert_array_equal(gpu.rssi, cpu.rssi)
        assert gpu.pkt_offset[-1] == gpu.count
        assert engine.result_count() == (cpu.count, 0)
    src_dev.free()
    rec_dev.free()


def test_generator_walk_over_thousands_of_packets(engine, rsa, O):
    """k_rng_chain walks java.util.Random over the packets a wave at a time, speculating that Tx draws
    succeed: 2500 frames in one tick (three blocks of 1024, chunks of 64) with every kind of packet --
    Tx probability 0 (no draw, receivers skipped), fractional at several failure rates, 1 with
    successRatioRx 1 elsewhere -- must leave every verdict and the generator state as the serial
    Java loop does, tick after tick."""
    n, t = 20000, 2500
    rng = np.random.default_rng(77)
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=78)
    nd.txprob[:] = rng.choice([0.0, 0.05, 0.3, 0.6, 0.97, 1.0], n, p=[0.05, 0.1, 0.2, 0.25, 0.3, 0.1])
    nd.rxprob[:] = rng.choice([0.0, 0.5, 0.9, 1.0], n, p=[0.02, 0.3, 0.3, 0.38])
    for ratio in (1.0, 0.85):
        params = dict(udgm_success_ratio_rx=ratio)
        configure_engine(engine, nd, "udgm", params)
        mdl = oracle_model(O, "udgm", params)
        engine.seed(2024)
        state = O.lib().orc_jrandom_seed(2024)
        for k in range(2):
            srcs = np.sort(rng.choice(n, t, replace=False)).astype(np.int32)
            pk = nd.packets(srcs, start_us=1000 * k, air_us=8128)
            cpu = O.tick(mdl, nd, pk, rng_state=state)
            state = cpu.rng_state
            gpu = engine.tick(to_tx_records(rsa, pk))
            assert_same(gpu, cpu, "ratio %.2f tick %d" % (ratio, k))
            assert engine.rng_state == state
            assert 0 < int(cpu.pkt_interference.sum()) < t


def test_device_records_with_a_fractional_tx_probability_are_refused_without_draws(engine, rsa, O):
    """Records the caller builds in device memory are not inspected by the host: when no node probability asks for
    draws, a record whose txprob is strictly between 0 and 1 cannot be honoured (the reference would draw for it,
    UDGMRadioMedium.java:87-92) -- the kernels flag it and the result reads as RM_ERR_STATE instead of being silently
    wrong; with a lossy node in the table the draw kernels run and the same records are exact."""
    from util import DeviceArray
    n = 3000
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=45)
    srcs = np.arange(0, n, 60)
    pk = nd.packets(srcs, 0, 320)
    pk["txprob"][3] = 0.5
    recs = to_tx_records(rsa, pk)
    for kind in ("udgm", "logdist"):
        configure_engine(engine, nd, kind, {})
        dev = DeviceArray(recs)
        engine.tick_run_device(0, 1000, dev.ptr.value, len(recs))
        with pytest.raises(rsa.RadioMediumError) as e:
            engine.result_copy(len(recs))
        assert e.value.code == -5 and "txprob" in str(e.value)
        # the same tick with its records' probabilities as the table has them: fine
        pk_ok = nd.packets(srcs, 0, 320)
        dev2 = DeviceArray(to_tx_records(rsa, pk_ok))
        engine.tick_run_device(0, 1000, dev2.ptr.value, len(recs))
        gpu = engine.result_copy(len(recs))
        cpu = O.tick(oracle_model(O, kind, {}), nd, pk_ok)
        np.testing.assert_array_equal(gpu.dst, cpu.dst)
        np.testing.assert_array_equal(gpu.verdict, cpu.verdict)
        dev.free()
        dev2.free()
    # a lossy receiver somewhere: the draw kernels run, the record's own probability is honoured
    nd.rxprob[7] = 0.5
    configure_engine(engine, nd, "udgm", {})
    engine.seed(11)
    dev = DeviceArray(recs)
    engine.tick_run_device(0, 1000, dev.ptr.value, len(recs))
    gpu = engine.result_copy(len(recs))
    cpu = O.tick(oracle_model(O, "udgm", {}), nd, pk, rng_state=O.lib().orc_jrandom_seed(11))
    np.testing.assert_array_equal(gpu.dst, cpu.dst)
    np.testing.assert_array_equal(gpu.verdict, cpu.verdict)
    assert engine.rng_state == cpu.rng_state
    dev.free()
