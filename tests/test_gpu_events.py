"""The reception stage on the device against the oracle's serial replay of the
reference's event path (oracle/rm_events.c: the ladder queue of com/botbox/scheduler/EventQueue.java, literal;
Simulator.generate*Events / processAllEvents; ReceptionEvent / TransmissionEvent.execute; Transciever state):
after every tick the delivery list -- the Simulator.deliverRadioPacket calls of the drain, in call order, equal
timestamps included -- and every node's (rssi, receiving state) must be identical."""
import numpy as np
import pytest

from util import DeviceArray, KINDS, configure_engine, oracle_model, random_nodes, to_tx_records

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fresh_reports(rsa, monkeypatch):
    """rm_events_enable begins a new session: the engine has reported nothing of it yet (check_drain's shadow of the reports)"""
    orig = rsa.Engine.events_enable

    def enable(self, *a, **kw):
        self._reported = None
        return orig(self, *a, **kw)
    monkeypatch.setattr(rsa.Engine, "events_enable", enable)


def oracle_deliveries(O, ev):
    d = ev[ev["kind"] == O.EV_RX_END_DELIVERY]
    return d["pkt"].astype(np.int64), d["node"].astype(np.int32), d["rssi"].astype(np.float64)


def check_drain(O, eng, sim, t, nodes, what, immediate=(), own=None):
    ev = sim.step(t)
    pkt, dst, rssi = oracle_deliveries(O, ev)
    if immediate:     # the constant-loss medium delivered synchronously, before the drain
        pkt = np.concatenate([np.array([i[0] for i in immediate], dtype=np.int64), pkt])
        dst = np.concatenate([np.array([i[1] for i in immediate], dtype=np.int32), dst])
        rssi = np.concatenate([np.array([i[2] for i in immediate]), rssi])
    if own is not None:
        keep = (dst >= own[0]) & (dst < own[0] + own[1])
        pkt, dst, rssi = pkt[keep], dst[keep], rssi[keep]
    gp, gd, gr, pending = eng.events_process(t)
    assert len(gp) == len(pkt), "%s: %d deliveries, oracle %d" % (what, len(gp), len(pkt))
    np.testing.assert_array_equal(gp, pkt, err_msg=what + " packet order")
    np.testing.assert_array_equal(gd, dst, err_msg=what + " destination order")
    np.testing.assert_array_equal(gr, rssi, err_msg=what + " rssi")
    want_rssi, want_state = sim.node_info(enabled=nodes.enabled)
    got_rssi, got_state, got_ch = eng.node_info()
    sel = slice(None) if own is None else slice(own[0], own[0] + own[1])
    np.testing.assert_array_equal(got_state[sel], want_state[sel], err_msg=what + " receiving state")
    np.testing.assert_array_equal(got_rssi[sel], want_rssi[sel], err_msg=what + " rssi of node-info")
    np.testing.assert_array_equal(got_ch, nodes.channel, err_msg=what + " channel")
    # ... and incrementally (rm_node_info_changed): the nodes reported since the last call, applied to what was reported before,
    # give the same table; the first call reports every node, later ones only nodes whose fields really differ
    cn, cr, cs, cc = eng.node_info_changed()
    shadow = getattr(eng, "_reported", None)
    if shadow is None or len(shadow[0]) != len(got_rssi):
        assert sorted(cn.tolist()) == list(range(len(got_rssi))), what + ": a first report names every node once"
        shadow = [np.zeros(len(got_rssi)), np.zeros(len(got_rssi), dtype=np.int32), np.zeros(len(got_rssi), dtype=np.int32)]
    else:
        assert len(set(cn.tolist())) == len(cn), what + ": a node reported twice"
        same = (shadow[0][cn].view(np.int64) == cr.view(np.int64)) & (shadow[1][cn] == cs) & (shadow[2][cn] == cc)
        assert not same.any(), what + ": reported without a change"
    shadow[0][cn], shadow[1][cn], shadow[2][cn] = cr, cs, cc
    eng._reported = shadow
    np.testing.assert_array_equal(shadow[0].view(np.int64), got_rssi.view(np.int64), err_msg=what + " incremental rssi")
    np.testing.assert_array_equal(shadow[1], got_state, err_msg=what + " incremental state")
    np.testing.assert_array_equal(shadow[2], got_ch, err_msg=what + " incremental channel")
    return len(pkt)


def random_session(O, rsa, eng, seed, kind, params, n=1500, ticks=14, per_tick=30, tick_styles=(1000,), aligned=False,
                   hex_lengths=(0, 2, 20, 64, 254), matrix=None, on_air=False):
    rng = np.random.default_rng(seed)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nodes = random_nodes(O, n, side, seed)
    nodes.rxprob[rng.choice(n, n // 5, replace=False)] = 0.6
    nodes.txprob[rng.choice(n, n // 20, replace=False)] = 0.5
    nodes.enabled[rng.choice(n, n // 50, replace=False)] = 0
    configure_engine(eng, nodes, kind, params, matrix)
    eng.seed(seed)
    eng.events_enable()
    eng.set_time(0)        # a new Simulator starts at time 0 (the engine may have served another session)
    mdl = oracle_model(O, kind, params, matrix)
    state = O.lib().orc_jrandom_seed(seed)
    sim = O.Sim(n)
    now, base, delivered = 0, 0, 0
    onair = np.zeros(0, dtype=O.PACKET_DTYPE)      # (on_air: the SINR extension's frames of earlier ticks, rm_tick_begin's rule)
    for k in range(ticks):
        t_end = now + int(rng.choice(tick_styles))
        t = int(rng.integers(0, per_tick + 1))
        src = rng.choice(n, t, replace=False)
        pk = nodes.packets(src, 0, 0)
        pk["start_us"] = now if aligned else rng.integers(now - 50, t_end, t)      # a start before "now" is clamped
        pk["air_us"] = 32 * rng.choice(hex_lengths, t)
        assert eng.events_next_packet() == base
        got = eng.tick(to_tx_records(rsa, pk), now, t_end)
        if on_air:
            onair = onair[onair["start_us"] + onair["air_us"] > now]
            want = O.tick(mdl, nodes, np.concatenate([onair, pk]), first_new=len(onair), rng_state=state)
            onair = np.concatenate([onair, pk])
        else:
            want = O.tick(mdl, nodes, pk, rng_state=state)
        state = want.rng_state
        assert got.count == want.count
        imm = sim.medium_calls(want, pk, pkt_base=base, const_loss=(kind == "udgm_const"))
        base += t
        delivered += check_drain(O, eng, sim, t_end, nodes, "seed %d tick %d" % (seed, k), immediate=imm)
        now = t_end
    delivered += check_drain(O, eng, sim, now + 10 ** 7, nodes, "seed %d final drain" % seed)
    assert sim.pending == 0
    sim.close()
    eng.events_disable()
    return delivered


@pytest.mark.parametrize("seed", range(6))
def test_udgm_sessions_with_draws_random_starts(O, rsa, engine, seed):
    n = random_session(O, rsa, engine, 100 + seed, "udgm", dict(udgm_success_ratio_rx=0.8), tick_styles=(1000, 1000, 10, 3000))
    assert n > 0


@pytest.mark.parametrize("seed", range(4))
def test_aligned_frames_tie_everywhere(O, rsa, engine, seed):
    """the synthetic workload's shape: every frame of a tick starts at the tick's start and has the same air time --
    all starts tie, all ends tie, and a receiver of two frames sees tied events on itself"""
    n = random_session(O, rsa, engine, 200 + seed, "udgm", {}, aligned=True, hex_lengths=(254,), per_tick=60, ticks=20)
    assert n > 0
    n = random_session(O, rsa, engine, 300 + seed, "udgm", {}, aligned=True, hex_lengths=(0, 0, 62), per_tick=60)
    assert n > 0


@pytest.mark.parametrize("seed", range(3))
def test_sinr_sessions_with_frames_on_the_air(O, rsa, engine, seed):
    """the SINR extension in the closed loop: the tick's one-launch form leaves the on-air entries, the second launch decides
    capture and half duplex, and the reception stage takes the verdicts from the frames' segments"""
    n = random_session(O, rsa, engine, 700 + seed, "logdist", dict(ld_flags=1, ld_sigma_db=4.0, ld_seed=5 + seed), n=2500, per_tick=40,
                       ticks=16, hex_lengths=(10, 64, 254, 254), on_air=True)
    assert n > 0


def test_null_and_n2n_media(O, rsa, engine):
    rng = np.random.default_rng(5)
    assert random_session(O, rsa, engine, 400, "null", {}, n=300, per_tick=6) > 0
    m = rng.uniform(0, 1, (300, 300))
    assert random_session(O, rsa, engine, 401, "n2n", {}, n=300, per_tick=6, matrix=m) > 0


def test_constant_loss_delivers_synchronously(O, rsa, engine):
    assert random_session(O, rsa, engine, 500, "udgm_const", {}, n=1500) > 0


def test_logdist_sessions(O, rsa, engine):
    assert random_session(O, rsa, engine, 600, "logdist", dict(ld_sigma_db=4.0, ld_seed=77), n=3000, per_tick=40) > 0


def test_per_packet_transmit_feeds_the_event_stage(O, rsa, engine):
    """rm_transmit, the per-packet drop-in call, with the reception stage on: one packet per call, drained per tick"""
    n = 1200
    rng = np.random.default_rng(9)
    nodes = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), 9)
    configure_engine(engine, nodes, "udgm", {})
    engine.events_enable()
    mdl = oracle_model(O, "udgm", {})
    sim = O.Sim(n)
    base, now = 0, 0
    for k in range(6):
        for _ in range(5):
            s = int(rng.integers(0, n))
            start = now + int(rng.integers(0, 1000))
            hexlen = int(rng.choice([10, 254]))
            pk = nodes.packets([s], start, 32 * hexlen)
            want = O.tick(mdl, nodes, pk)
            got = engine.transmit(s, start, hexlen)
            assert got.count == want.count
            sim.medium_calls(want, pk, pkt_base=base)
            base += 1
        now += 1000
        check_drain(O, engine, sim, now, nodes, "transmit tick %d" % k)
    check_drain(O, engine, sim, now + 10 ** 6, nodes, "transmit final")
    sim.close()


def shaped_run(O, rsa, eng, n, t, cfg_index, model_kw, ticks, okw):
    """BASELINE-shaped ticks through the device-resident path (source indices in HBM): frames of 8128 us over
    1000 us ticks, so ~9 ticks of frames are pending at any time"""
    from radio_sim_amd import workload as W
    nodes_w = W.make_nodes(n, cfg_index)
    nodes = O.NodeTable(n)
    nodes.x, nodes.y = nodes_w.x, nodes_w.y
    eng.upload_table(nodes)
    eng.set_model(rsa.MODEL_LOGDIST, **model_kw)
    eng.events_enable(1 << 15, 1 << 20)
    mdl = O.model(O.MODEL_LOGDIST, **okw)
    sim = O.Sim(n)
    base, total = 0, 0
    for k in range(ticks):
        src = W.choose_sources(n, t, 0xC0FFEE00 + cfg_index, k)
        dev = DeviceArray(src)
        now = k * W.TICK_US
        eng.tick_run_sources_device(now, now + W.TICK_US, dev.ptr.value, t, now, W.AIR_US)
        pk = nodes.packets(src, now, W.AIR_US)
        want = O.tick(mdl, nodes, pk)
        sim.medium_calls(want, pk, pkt_base=base)
        base += t
        total += check_drain(O, eng, sim, now + W.TICK_US, nodes, "tick %d" % k)
        dev.free()
    total += check_drain(O, eng, sim, (ticks + 20) * W.TICK_US, nodes, "final drain")
    assert sim.pending == 0
    sim.close()
    return total


def test_c2_shape_ten_thousand_nodes(O, rsa, engine):
    total = shaped_run(O, rsa, engine, 10_000, 100, 2, {}, 14, {})
    assert total > 10_000


def test_c3_shape_hundred_thousand_nodes_shadowing(O, rsa, engine):
    kw = dict(ld_sigma_db=4.0, ld_seed=0xC0FFEE)
    total = shaped_run(O, rsa, engine, 100_000, 1000, 3, kw, 11, kw)
    assert total > 100_000


def test_receiver_partitions_keep_their_own_events(O, rsa):
    """two contexts, each a receiver range partition: every context's delivery list is the oracle's list filtered to
    its receivers (same global order), its node states those of its nodes; the transmission events of a packet
    live where its source does"""
    n = 2000
    rng = np.random.default_rng(21)
    nodes = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), 21)
    parts = [(0, 900), (900, 1100)]
    engs = []
    for first, count in parts:
        e = rsa.Engine(0)
        configure_engine(e, nodes, "udgm", {})
        e.set_partition(first, count)
        e.events_enable()
        engs.append(e)
    mdl = oracle_model(O, "udgm", {})
    sims = [O.Sim(n) for _ in parts]    # the oracle replay per rank = the global replay (same events everywhere)
    base, now = 0, 0
    try:
        for k in range(10):
            t = int(rng.integers(1, 40))
            src = rng.choice(n, t, replace=False)
            pk = nodes.packets(src, 0, 0)
            pk["start_us"] = rng.integers(now, now + 1000, t)
            pk["air_us"] = 32 * rng.choice([0, 20, 254], t)
            want = O.tick(mdl, nodes, pk)
            for e, sim, own in zip(engs, sims, parts):
                got = e.tick(to_tx_records(rsa, pk), now, now + 1000)
                sim.medium_calls(want, pk, pkt_base=base)
                check_drain(O, e, sim, now + 1000, nodes, "rank %s tick %d" % (own, k), own=own)
            base += t
            now += 1000
        for e, sim, own in zip(engs, sims, parts):
            check_drain(O, e, sim, now + 10 ** 6, nodes, "rank %s final" % (own,), own=own)
    finally:
        for e in engs:
            e.close()
        for s in sims:
            s.close()
