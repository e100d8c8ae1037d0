"""Parity of the extension medium (log-distance path loss, log-normal shadowing, co-channel SINR
capture, multi-tick overlap -- DESIGN.md "Extension spec") with the CPU oracle's independent
implementation of the same text.  Verdicts / heard sets bit-exact; rssi and sinr are compared
bit-exactly as well (tolerance of the north star: 1e-5 relative)."""
import os

import numpy as np
import pytest

from util import (run_both, assert_same, random_nodes, to_tx_records, configure_engine, oracle_model, sinr_lists_forced)

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["scan", "lists"])
def sinr_form(request, monkeypatch):
    """Both forms of the SINR medium's lone tick: by scan (the interferers of a heard link found among the frames on the air,
    rm_airscan.hip -- the default wherever the one-launch tick applies) and with the per-receiver lists kept on the device
    (RM_SINR_SCAN=0: what larger ticks and unsorted tables take)."""
    forced = sinr_lists_forced()                 # (tools/knob_sweep.sh)
    if request.param == "lists":
        monkeypatch.setenv("RM_SINR_SCAN", "0")
    elif forced:
        pytest.skip("the run's knobs keep the lists")
    return request.param


def _layout(O, n, seed, z=0.0, k=20.0):
    side = 50.0 * np.sqrt(np.pi * n / k)
    return random_nodes(O, n, side, seed=seed, z_span=z)


@pytest.mark.parametrize("params", [
    {},
    {"ld_exponent": 2.0, "ld_pl0_db": 46.7, "ld_sensitivity_dbm": -90.0},
    {"ld_exponent": 3.5, "ld_d0": 2.5},
    {"ld_sigma_db": 4.0, "ld_seed": 12345},
    {"ld_sigma_db": 8.0, "ld_seed": 7, "ld_clip": 2.0, "ld_exponent": 2.7},
])
def test_logdist_models(engine, rsa, O, params):
    n, t = 4000, 150
    nd = _layout(O, n, seed=31, z=10.0)
    rng = np.random.default_rng(3)
    nd.txpower[:] = rng.choice([0.0, -3.0, -7.0, 3.0], n)
    nd.channel[rng.random(n) < 0.1] = 25
    nd.enabled[rng.random(n) < 0.03] = 0
    src = np.sort(rng.choice(n, t, replace=False))
    gpu, cpu = run_both(O, rsa, engine, nd, "logdist", params, nd.packets(src, 0, 8128))
    assert cpu.count > 1000
    assert_same(gpu, cpu, "logdist %s" % params)
    assert np.all(gpu.rssi >= params.get("ld_sensitivity_dbm", -95.0))


@pytest.mark.parametrize("t", [1, 7, 8, 9, 150, 333])
def test_lone_tick_with_its_frames_dealt_to_the_xcds_in_eighths(engine, rsa, O, monkeypatch, t):
    """The one-launch tick renames its workgroups (XCD x takes the x-th eighth of the frames, rm_tick.hip: xcd_slot; RM_TICK_XCD_MAP=0
    turns it off): every frame is still evaluated exactly once and lands in its own slot -- frame counts around the multiples of 8."""
    monkeypatch.delenv("RM_TICK_XCD_MAP", raising=False)
    n = 6000
    nd = _layout(O, n, seed=77)
    rng = np.random.default_rng(t)
    src = np.sort(rng.choice(n, t, replace=False))
    params = {"ld_sigma_db": 4.0, "ld_seed": 99}
    gpu, cpu = run_both(O, rsa, engine, nd, "logdist", params, nd.packets(src, 0, 8128))
    assert cpu.count > 5 * t
    assert_same(gpu, cpu, "xcd map, %d frames" % t)


def test_logdist_tx_power_override_and_probabilities(engine, rsa, O):
    n = 3000
    nd = _layout(O, n, seed=5)
    rng = np.random.default_rng(9)
    nd.rxprob[:] = np.where(rng.random(n) < 0.6, 1.0, rng.uniform(0, 1, n))
    nd.rxprob[::50] = 0.0
    nd.txprob[:] = np.where(rng.random(n) < 0.7, 1.0, rng.uniform(0, 1, n))
    src = np.sort(rng.choice(n, 80, replace=False))
    pk = nd.packets(src, 0, 8128)
    pk["txpower"] = rng.uniform(-20, 10, len(pk))     # "rf-power" override per packet
    pk["txpower"][3] = -200.0                          # nobody can hear this one
    gpu, cpu = run_both(O, rsa, engine, nd, "logdist", {"ld_sigma_db": 3.0, "ld_seed": 1}, pk, seed=99)
    assert cpu.pkt_draws.sum() > 100
    assert_same(gpu, cpu, "logdist stochastic")
    assert engine.rng_state == cpu.rng_state


def test_shadowing_is_symmetric_and_seeded(engine, rsa, O):
    nd = _layout(O, 600, seed=8)
    p = {"ld_sigma_db": 6.0, "ld_seed": 42}
    configure_engine(engine, nd, "logdist", p)
    a = engine.tick(to_tx_records(rsa, nd.packets([10])))
    b = engine.tick(to_tx_records(rsa, nd.packets(a.dst[:5])))
    # link (10 -> j) and (j -> 10) see the same shadowing deviate (equal tx power): equal rssi
    for q, j in enumerate(a.dst[:5]):
        back = b.rssi[(b.pkt == q) & (b.dst == 10)]
        assert len(back) == 1 and back[0] == a.rssi[q]
    configure_engine(engine, nd, "logdist", {"ld_sigma_db": 6.0, "ld_seed": 43})
    c = engine.tick(to_tx_records(rsa, nd.packets([10])))
    assert not (len(c.rssi) == len(a.rssi) and np.array_equal(c.rssi, a.rssi))


def _sinr_params(**kw):
    p = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 77}
    p.update(kw)
    return p


def test_sinr_single_tick_16_channels(engine, rsa, O, sinr_form):
    n, t = 3000, 400          # dense traffic: many co-channel collisions
    nd = _layout(O, n, seed=21)
    rng = np.random.default_rng(4)
    nd.channel[:] = 11 + rng.integers(0, 16, n)
    src = np.sort(rng.choice(n, t, replace=False))
    pk = nd.packets(src, 0, 8128)
    pk["start_us"] = rng.integers(0, 1000, t)
    gpu, cpu = run_both(O, rsa, engine, nd, "logdist", _sinr_params(), pk)
    assert cpu.count > 500
    assert (cpu.verdict == O.INTERFERED).sum() > 20 and (cpu.verdict == O.DELIVERED).sum() > 20
    assert_same(gpu, cpu, "sinr 16ch")


def test_sinr_single_channel_heavy_interference(engine, rsa, O, sinr_form):
    n, t = 2000, 300
    nd = _layout(O, n, seed=22)
    rng = np.random.default_rng(5)
    src = np.sort(rng.choice(n, t, replace=False))
    pk = nd.packets(src, 0, 4000)
    pk["start_us"] = rng.integers(0, 3000, t)
    pk["air_us"] = rng.choice([320, 1280, 4064, 8128], t)       # short frames may not overlap
    gpu, cpu = run_both(O, rsa, engine, nd, "logdist", _sinr_params(ld_sigma_db=0.0, ld_capture_db=6.0), pk)
    assert (cpu.verdict == O.INTERFERED).sum() > 100
    assert_same(gpu, cpu, "sinr 1ch")


def test_sinr_multi_tick_overlap_and_half_duplex(engine, rsa, O, sinr_form):
    """Frames stay on the air over several ticks (8128 us frames, 1000 us ticks)."""
    n = 2500
    nd = _layout(O, n, seed=23)
    rng = np.random.default_rng(6)
    params = _sinr_params()
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    onair = np.zeros(0, dtype=O.PACKET_DTYPE)
    total_interfered = 0
    for tick in range(12):
        t0 = tick * 1000
        onair = onair[onair["start_us"] + onair["air_us"] > t0]            # rm_tick_begin pruning rule
        src = np.sort(rng.choice(n, 40, replace=False))
        if tick == 3:
            src[0] = int(onair["src"][0])                                    # a node transmits again while on air
            src = np.sort(src)
        new = nd.packets(src, 0, 0)
        new["start_us"] = t0 + rng.integers(0, 1000, len(new))
        new["air_us"] = rng.choice([320, 2048, 8128], len(new))
        active = np.concatenate([onair, new])
        cpu = O.tick(mdl, nd, active, first_new=len(onair))
        engine.tick_begin(t0, t0 + 1000)
        engine.enqueue_records(to_tx_records(rsa, new))
        gpu = engine.tick_flush()
        assert_same(gpu, cpu, "overlap tick %d" % tick)
        total_interfered += int((cpu.verdict == O.INTERFERED).sum())
        onair = active
    assert total_interfered > 50


def test_sinr_receiver_is_transmitting(engine, rsa, O, sinr_form):
    """Half duplex: a node that is itself on the air cannot receive an overlapping frame."""
    nd = O.NodeTable(3)
    nd.x[:] = [0.0, 20.0, 40.0]
    params = {"ld_flags": 1}
    pk = nd.packets([0, 1], 0, 1000)
    gpu, cpu = run_both(O, rsa, engine, nd, "logdist", params, pk)
    assert_same(gpu, cpu, "half duplex")
    # node 1 hears node 0 but transmits itself -> interfered; node 2 hears both -> collision
    v = {(p, d): x for p, d, x in zip(gpu.pkt, gpu.dst, gpu.verdict)}
    assert v[(0, 1)] == rsa.INTERFERED and v[(1, 0)] == rsa.INTERFERED
    pk2 = nd.packets([0, 1], 0, 1000)
    pk2["start_us"] = [0, 5000]                                                # no time overlap
    gpu, cpu = run_both(O, rsa, engine, nd, "logdist", params, pk2)
    assert_same(gpu, cpu, "no overlap")
    assert set(gpu.verdict.tolist()) == {rsa.DELIVERED}
    assert np.all(gpu.sinr > 10.0) and np.all(gpu.sinr == gpu.rssi - (-100.0))


def test_logdist_far_origin(engine, rsa, O):
    nd = _layout(O, 1500, seed=3)
    nd.x += 8.0e8
    nd.y += 8.0e8
    src = np.arange(0, 1500, 30)
    gpu, cpu = run_both(O, rsa, engine, nd, "logdist", {"ld_sigma_db": 4.0, "ld_seed": 5}, nd.packets(src))
    assert cpu.count > 500
    assert_same(gpu, cpu, "logdist far origin")


def test_sinr_device_resident_on_air_list(engine, rsa, O, sinr_form):
    """rm_tick_run_sources_device with the SINR medium: the frames of earlier ticks stay on the
    device as interferers (batches expire by start + air > t_begin, also out of order)."""
    from util import DeviceArray
    n = 3000
    nd = _layout(O, n, seed=29)
    rng = np.random.default_rng(8)
    nd.channel[:] = 11 + rng.integers(0, 2, n)
    params = _sinr_params()
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    airs = [8128, 2048, 320, 8128, 4064, 320, 2048, 8128, 8128, 320]
    onair = np.zeros(0, dtype=O.PACKET_DTYPE)
    interfered = 0
    for tick, air in enumerate(airs):
        t0 = tick * 1000
        onair = onair[onair["start_us"] + onair["air_us"] > t0]
        srcs = np.sort(rng.choice(n, 50, replace=False)).astype(np.int32)
        new = nd.packets(srcs, t0, air)
        active = np.concatenate([onair, new])
        cpu = O.tick(mdl, nd, active, first_new=len(onair))
        dev = DeviceArray(srcs)
        engine.tick_run_sources_device(t0, t0 + 1000, dev.ptr.value, len(srcs), t0, air)
        gpu = engine.result_copy(len(srcs))
        dev.free()
        assert_same(gpu, cpu, "device on-air list, tick %d" % tick)
        interfered += int((cpu.verdict == O.INTERFERED).sum())
        onair = active
    assert interfered > 100


def test_sinr_host_and_device_ticks_share_the_frames_on_the_air(engine, rsa, O, sinr_form):
    """One window of frames on the air per context, whoever brought them: ticks whose records come from the host
    (rm_tick_flush) and ticks built on the device from source indices (rm_tick_run_sources_device) interfere with each other."""
    from util import DeviceArray
    n = 4000
    nd = _layout(O, n, seed=31)
    rng = np.random.default_rng(9)
    params = _sinr_params()
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    onair = np.zeros(0, dtype=O.PACKET_DTYPE)
    interfered = 0
    for tick in range(14):
        t0 = tick * 1000
        onair = onair[onair["start_us"] + onair["air_us"] > t0]
        srcs = np.sort(rng.choice(n, 60, replace=False)).astype(np.int32)
        air = int(rng.choice([2048, 8128]))
        new = nd.packets(srcs, t0, air)
        cpu = O.tick(mdl, nd, np.concatenate([onair, new]), first_new=len(onair))
        if tick % 3 == 1:
            dev = DeviceArray(srcs)
            engine.tick_run_sources_device(t0, t0 + 1000, dev.ptr.value, len(srcs), t0, air)
            gpu = engine.result_copy(len(srcs))
            dev.free()
        else:
            engine.tick_begin(t0, t0 + 1000)
            engine.enqueue_records(to_tx_records(rsa, new))
            gpu = engine.tick_flush()
        assert_same(gpu, cpu, "shared window, tick %d" % tick)
        interfered += int((cpu.verdict == O.INTERFERED).sum())
        onair = np.concatenate([onair, new])
    assert interfered > 100


def _overlap_run(engine, rsa, O, nd, mdl, rng, ticks, per_tick, airs, hook=None, t_of=None, what=""):
    """ticks of new frames through the tick API against the oracle's tick over the full on-air list"""
    n = nd.n
    onair = np.zeros(0, dtype=O.PACKET_DTYPE)
    interfered = 0
    for tick in range(ticks):
        t0 = tick * 1000 if t_of is None else t_of(tick)
        if hook is not None:
            hook(tick)
        onair = onair[onair["start_us"] + onair["air_us"] > t0]
        k = int(per_tick(tick)) if callable(per_tick) else per_tick
        src = np.sort(rng.choice(n, k, replace=False)) if k else np.zeros(0, dtype=np.int64)
        new = nd.packets(src, 0, 0)
        new["start_us"] = t0 + rng.integers(0, 1000, len(new))
        new["air_us"] = rng.choice(airs, len(new))
        active = np.concatenate([onair, new])
        cpu = O.tick(mdl, nd, active, first_new=len(onair))
        engine.tick_begin(t0, t0 + 1000)
        engine.enqueue_records(to_tx_records(rsa, new))
        try:
            gpu = engine.tick_flush()
        except rsa.RadioMediumError as e:
            raise AssertionError("%s tick %d (%d new frames, %d on the air, lists %r): %s" % (
                what, tick, len(new), len(onair), engine.air_list_stats(), e))
        assert_same(gpu, cpu, "%s tick %d" % (what, tick))
        interfered += int((cpu.verdict == O.INTERFERED).sum())
        onair = active
    return interfered


def test_sinr_lists_live_across_ticks(engine, rsa, O, sinr_form):
    """The per-receiver interferer lists stay on the device: a tick adds its new frames only.  A small link capacity
    makes the entry rings wrap many times; frames of very different lengths leave the air out of order; empty ticks."""
    n = 10000                                       # enough receiver groups to spread the candidates over the shards
    nd = _layout(O, n, seed=41)
    rng = np.random.default_rng(12)
    params = _sinr_params()
    configure_engine(engine, nd, "logdist", params)
    # 2048 entries per sub-ring: ~1M entries wrap them twice (without the shadowing table -- a developer knob -- the
    # filter hands six times the candidates to the exact stage, and the candidate shards need the room instead)
    engine.set_link_capacity(1 << 19 if not os.environ.get("RM_NO_SHADOW_TABLE") else 1 << 23)
    mdl = oracle_model(O, "logdist", params)
    inc0, reb0 = engine.air_list_stats()
    sc0 = engine.air_scan_ticks()
    got = _overlap_run(engine, rsa, O, nd, mdl, rng, 200, lambda t: 0 if t % 17 == 5 else rng.integers(1, 30),
                       [320, 2048, 8128, 8128, 20000], what="rings")
    inc, reb = engine.air_list_stats()
    assert got > 100
    if sinr_form == "scan":                         # no lists at all (12 ticks had no frames)
        assert (inc - inc0, reb - reb0, engine.air_scan_ticks() - sc0) == (0, 0, 188)
    elif os.environ.get("RM_AIR_LISTS") != "0":      # (the developer knob that rebuilds the lists every tick)
        assert (inc - inc0, reb - reb0) == (187, 1) # one build, then only new frames (12 ticks had none)


def test_sinr_lists_rebuilt_when_something_changes(engine, rsa, O, sinr_form):
    """Whatever an old entry was computed from may change while its frame is on the air: a receiver moves or changes
    its channel, the model changes, the clock goes back.  The next tick rebuilds the lists from every frame on the
    air (the oracle evaluates the full on-air list against the node table as it is now)."""
    n = 2000
    nd = _layout(O, n, seed=43)
    rng = np.random.default_rng(13)
    params = _sinr_params()
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)

    def hook(tick):
        if tick in (4, 9):                          # a receiver in the middle of the traffic moves next to a sender
            i = int(rng.integers(n))
            nd.x[i], nd.y[i] = nd.x[(i + 1) % n] + 1.0, nd.y[(i + 1) % n]
            if tick == 9:
                nd.channel[i] = 11
            engine.update_node(i, nd.x[i], nd.y[i], nd.z[i], nd.txpower[i], int(nd.channel[i]), int(nd.enabled[i]),
                               nd.rxprob[i], nd.txprob[i])
        if tick == 14:
            engine.move_nodes(np.arange(5, dtype=np.int32), nd.x[:5] + 3.0, nd.y[:5])
            nd.x[:5] += 3.0
            nd.z[:5] = 0.0

    inc0, reb0 = engine.air_list_stats()
    # tick 12 begins before tick 11 did: frames that had left the air do not come back
    t_of = lambda t: (t * 1000 if t != 12 else 9500)
    got = _overlap_run(engine, rsa, O, nd, mdl, rng, 20, 35, [320, 2048, 8128], hook=hook, t_of=t_of, what="changes")
    inc, reb = engine.air_list_stats()
    assert got > 30
    if sinr_form == "scan":                         # nothing to rebuild: every tick looks at the table as it is
        assert (inc - inc0, reb - reb0) == (0, 0)
    elif os.environ.get("RM_AIR_LISTS") != "0":
        assert reb - reb0 == 5 and inc - inc0 == 15 # first tick, three node changes, the clock going back


def test_sinr_lists_after_a_dropped_tick(engine, rsa, O, sinr_form):
    """A tick whose links do not fit the capacity is reported and leaves the lists unusable; with more room the
    next tick rebuilds them and is exact again."""
    n = 10000
    nd = _layout(O, n, seed=47)
    rng = np.random.default_rng(14)
    params = _sinr_params()
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    engine.set_link_capacity(1 << 16)
    onair = np.zeros(0, dtype=O.PACKET_DTYPE)
    failed = 0
    for tick in range(8):
        t0 = tick * 1000
        onair = onair[onair["start_us"] + onair["air_us"] > t0]
        src = np.sort(rng.choice(n, 900 if tick == 3 else 10, replace=False))
        new = nd.packets(src, t0, 8128)
        active = np.concatenate([onair, new])
        cpu = O.tick(mdl, nd, active, first_new=len(onair))
        engine.tick_begin(t0, t0 + 1000)
        engine.enqueue_records(to_tx_records(rsa, new))
        try:
            gpu = engine.tick_flush()
            assert_same(gpu, cpu, "after a dropped tick, tick %d" % tick)
        except rsa.RadioMediumError as e:
            assert "capacity" in str(e)
            failed += 1
            engine.set_link_capacity(1 << 22)
        onair = active
    assert failed == 1


def test_sinr_scan_through_the_frame_grid(engine, rsa, O):
    """The tick by scan looks its frames up in a 64 x 64 grid when their reach is small against the world (rm_airscan.hip):
    weak transmitters on a large field, thousands of frames on the air; a tight cluster of transmitters overfills its cell
    (more than 16 frames: the rest go to the list every new frame looks at); one frame has no bound at all (its transmitter
    shouts), one transmitter moves while its frame is on the air (half duplex goes by node, not by place)."""
    if sinr_lists_forced():
        pytest.skip("the run's knobs keep the lists")
    n = 20000
    nd = _layout(O, n, seed=61)
    rng = np.random.default_rng(16)
    nd.txpower[:] = -20.0
    cluster = np.arange(100, 160)
    nd.x[cluster] = nd.x[100] + rng.uniform(0, 8, len(cluster))
    nd.y[cluster] = nd.y[100] + rng.uniform(0, 8, len(cluster))
    nd.txpower[7] = 40.0
    params = _sinr_params()
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    onair = np.zeros(0, dtype=O.PACKET_DTYPE)
    interfered = deaf = 0
    sc0 = engine.air_scan_ticks()
    for tick in range(9):
        t0 = tick * 1000
        if tick == 6:                               # a node with a frame on the air is somewhere else now
            j = int(onair["src"][-1])
            nd.x[j], nd.y[j] = nd.x[(j + 11) % n] + 1.0, nd.y[(j + 11) % n]
            engine.update_node(j, nd.x[j], nd.y[j], nd.z[j], nd.txpower[j], int(nd.channel[j]), 1, 1.0, 1.0)
        onair = onair[onair["start_us"] + onair["air_us"] > t0]
        src = rng.choice(np.setdiff1d(np.arange(n), cluster), 340, replace=False)
        src = np.sort(np.concatenate([src, cluster[rng.random(len(cluster)) < 0.5], [7] if tick == 3 else []]).astype(np.int64))
        src = np.unique(src)
        new = nd.packets(src, 0, 0)
        new["start_us"] = t0 + rng.integers(0, 1000, len(new))
        new["air_us"] = rng.choice([2048, 8128, 8128], len(new))
        active = np.concatenate([onair, new])
        cpu = O.tick(mdl, nd, active, first_new=len(onair))
        engine.tick_begin(t0, t0 + 1000)
        engine.enqueue_records(to_tx_records(rsa, new))
        gpu = engine.tick_flush()
        assert_same(gpu, cpu, "frame grid, tick %d (%d frames on the air)" % (tick, len(active)))
        interfered += int((cpu.verdict == O.INTERFERED).sum())
        onair = active
    assert interfered > 200 and engine.air_scan_ticks() - sc0 == 9


def test_sinr_scan_with_more_near_frames_than_its_list_holds(engine, rsa, O):
    """1500 co-channel frames within reach of each other: a wave's quarter of the near list (256) overflows and the new
    frame's workgroup goes over the frames once more, pair phase by pair phase."""
    n, t = 3000, 1500
    nd = _layout(O, n, seed=67)
    rng = np.random.default_rng(17)
    src = np.sort(rng.choice(n, t, replace=False))
    pk = nd.packets(src, 0, 8128)
    pk["start_us"] = rng.integers(0, 1000, t)
    gpu, cpu = run_both(O, rsa, engine, nd, "logdist", _sinr_params(), pk)
    assert cpu.count > 5000 and (cpu.verdict == O.INTERFERED).sum() > 1000
    assert_same(gpu, cpu, "near list overflow")


def test_sinr_lists_with_the_per_frame_candidate_kernel(engine, rsa, O, monkeypatch):
    """The three-kernel form of a lone SINR tick (rebuild ticks, more than 4096 frames): tables large enough for the tiled
    filter take their candidates from k_frames_cand (one frame per workgroup); RM_FILTER=wg selects that regime for a
    table of test size."""
    monkeypatch.setenv("RM_FILTER", "wg")
    monkeypatch.setenv("RM_SINR_FRAMES", "0")       # (the one-launch form of the SINR tick would take these ticks otherwise)
    n = 6000
    nd = _layout(O, n, seed=53, z=3.0)
    rng = np.random.default_rng(15)
    nd.channel[:] = 11 + rng.integers(0, 3, n)
    params = _sinr_params()
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)

    def hook(tick):
        if tick == 7:                                   # a rebuild in the middle: every frame on the air is swept again
            engine.move_nodes(np.arange(3, dtype=np.int32), nd.x[:3] + 1.0, nd.y[:3], nd.z[:3])
            nd.x[:3] += 1.0

    got = _overlap_run(engine, rsa, O, nd, mdl, rng, 16, lambda t: rng.integers(0, 60), [320, 2048, 8128], hook=hook, what="frames cand")
    assert got > 30
