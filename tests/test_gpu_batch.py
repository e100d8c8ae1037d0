"""rm_batch_*: several independent ticks in one launch sequence.  Every tick of a batch must be what
the oracle (and a single rm_tick_run_sources_device call) gives for it -- including the order in
which the shared java.util.Random is consumed from tick to tick."""
import numpy as np
import pytest

from util import configure_engine, oracle_model, to_tx_records, random_nodes, assert_same, DeviceArray

pytestmark = pytest.mark.gpu

AIR = 8128


@pytest.fixture(autouse=True, params=[None, "2"], ids=["", "near-lists"])
def near_lists(request, monkeypatch):
    """Every case twice: as the sizes here send it, and with the near-frame lists of the batch filter (k_near_lists: phase A of a
    filter workgroup looks at the frames near its block of 16 workgroups) taken at any size -- by themselves they begin at 64
    workgroups and 512 frames per tick (tests/test_gpu_fullsize.py runs there)."""
    if request.param:
        monkeypatch.setenv("RM_NEAR_LISTS", request.param)
        monkeypatch.setenv("RM_WG_RPT", "4")      # (the lists go with the filter's 1024-receiver workgroups: taken here at any size, too)


CASES = [
    ("udgm", {}, False),                                                    # reference UDGM, no draws
    ("udgm", dict(udgm_success_ratio_rx=0.8), True),                        # every heard link draws
    ("udgm_const", {}, False),
    ("logdist", dict(ld_sigma_db=0.0), False),
    ("logdist", dict(ld_sigma_db=4.0, ld_seed=11), False),                  # shadowing table in the filter
]


def _layout(O, n, seed, lossy=False):
    nd = random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=seed)
    if lossy:
        rng = np.random.default_rng(seed + 100)
        nd.rxprob[rng.choice(n, n // 5, replace=False)] = 0.6
        nd.txprob[rng.choice(n, n // 50, replace=False)] = 0.5
        nd.enabled[rng.choice(n, n // 40, replace=False)] = 0
    return nd


def _ticks(n, n_ticks, per_tick, seed, ragged=False):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(n_ticks):
        t = per_tick if not ragged else max(1, per_tick - 11 * b)
        out.append(np.sort(rng.choice(n, t, replace=False)).astype(np.int32))
    return out


@pytest.mark.parametrize("kind,params,lossy", CASES)
@pytest.mark.parametrize("n_ticks", [1, 3, 6, 13])
def test_batch_matches_the_oracle_tick_by_tick(engine, rsa, O, kind, params, lossy, n_ticks):
    n = 6000
    nd = _layout(O, n, seed=5, lossy=lossy)
    configure_engine(engine, nd, kind, params)
    engine.seed(77)
    state = O.lib().orc_jrandom_seed(77)
    mdl = oracle_model(O, kind, params)
    srcs = _ticks(n, n_ticks, 150, seed=n_ticks, ragged=True)
    dev = [DeviceArray(s) for s in srcs]
    starts = [1000 * b for b in range(n_ticks)]
    engine.batch_run_sources_device(starts, [s + 1000 for s in starts], [d.ptr.value for d in dev],
                                    [len(s) for s in srcs], starts, [AIR] * n_ticks)
    views, status = engine.batch_result_view(n_ticks)      # all slots through the pinned host block, one launch
    assert status == [0] * n_ticks
    for b in range(n_ticks):
        pk = nd.packets(srcs[b], start_us=starts[b], air_us=AIR)
        cpu = O.tick(mdl, nd, pk, rng_state=state)
        state = cpu.rng_state
        gpu = engine.batch_result_copy(b, len(srcs[b]))
        assert cpu.count > 0
        assert_same(gpu, cpu, "%s tick %d of %d" % (kind, b, n_ticks))
        assert_same(views[b], cpu, "%s tick %d of %d in the host block" % (kind, b, n_ticks))
        np.testing.assert_array_equal(views[b].pkt_interference, cpu.pkt_interference)
        np.testing.assert_array_equal(views[b].pkt_offset, gpu.pkt_offset)
        assert engine.batch_result_count(b) == (cpu.count, 0)
    assert engine.rng_state == state
    for d in dev:
        d.free()


def test_batch_equals_single_ticks_and_slot0_is_the_plain_result(engine, rsa, O, near_lists, request):
    n = 20000
    nd = _layout(O, n, seed=9, lossy=True)
    params = dict(udgm_success_ratio_rx=0.9)
    configure_engine(engine, nd, "udgm", params)
    engine.profile_enable(1)
    srcs = _ticks(n, 5, 400, seed=2)
    dev = [DeviceArray(s) for s in srcs]
    starts = [1000 * b for b in range(5)]
    engine.seed(5)
    singles = []
    for b in range(5):
        engine.tick_run_sources_device(starts[b], starts[b] + 1000, dev[b].ptr.value, len(srcs[b]), starts[b], AIR)
        singles.append(engine.result_copy(len(srcs[b])))
    after_singles = engine.rng_state
    engine.seed(5)
    engine.batch_run_sources_device(starts, [s + 1000 for s in starts], [d.ptr.value for d in dev], [400] * 5, starts, [AIR] * 5)
    for b in range(5):
        assert_same(engine.batch_result_copy(b, 400), singles[b], "tick %d" % b)
    assert engine.rng_state == after_singles
    # (the second parametrisation does what it says: the batch's filter went through the near-frame lists)
    import os
    assert any(k.startswith("k_near_lists") for k in engine.profile_kernels()) == ("near-lists" in request.node.name or os.environ.get("RM_NEAR_LISTS") == "2")
    assert_same(engine.result_copy(400), singles[0], "slot 0 through rm_result_copy")
    # a plain tick after a batch reuses slot 0 and is unaffected by the other slots
    engine.seed(5)
    engine.tick_run_sources_device(0, 1000, dev[0].ptr.value, 400, 0, AIR)
    assert_same(engine.result_copy(400), singles[0], "single tick after a batch")
    assert_same(engine.batch_result_copy(3, 400), singles[3], "slot 3 still holds its tick")
    for d in dev:
        d.free()


def test_batch_of_given_records_and_repeated_batches(engine, rsa, O):
    """rm_batch_run_device (records given) + the parity double buffers over several batches."""
    n = 8000
    nd = _layout(O, n, seed=21)
    params = dict(ld_sigma_db=3.0, ld_seed=4)
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    rng = np.random.default_rng(8)
    for rep in range(3):
        n_ticks = 4 - rep
        srcs = _ticks(n, n_ticks, 120 + 50 * rep, seed=30 + rep)
        pks = [nd.packets(s, start_us=1000 * b, air_us=AIR) for b, s in enumerate(srcs)]
        for pk in pks:                                            # per-packet overrides (rf-power, channel)
            pk["txpower"] = rng.uniform(-3.0, 0.0, len(pk))
        dev = [DeviceArray(to_tx_records(rsa, pk)) for pk in pks]
        tb = [1000 * b for b in range(n_ticks)]
        engine.batch_run_device(tb, [t + 1000 for t in tb], [d.ptr.value for d in dev], [len(pk) for pk in pks])
        for b in range(n_ticks):
            assert_same(engine.batch_result_copy(b, len(pks[b])), O.tick(mdl, nd, pks[b]), "batch %d tick %d" % (rep, b))
        for d in dev:
            d.free()


def test_batch_falls_back_to_single_sequences_where_the_batched_kernels_do_not_apply(engine, rsa, O):
    """Coordinates of 1e6 m: the fp32 frame is too coarse (fp64 filter), no batched kernels -- the
    same API still returns every tick's result; an empty tick in the batch does too."""
    n = 3000
    nd = random_nodes(O, n, 400.0, seed=3)
    nd.x += 1.0e9
    configure_engine(engine, nd, "udgm", {})
    mdl = oracle_model(O, "udgm", {})
    srcs = _ticks(n, 3, 40, seed=1)
    srcs[1] = srcs[1][:0]
    dev = [DeviceArray(s) if len(s) else DeviceArray(nbytes=4) for s in srcs]
    tb = [0, 1000, 2000]
    engine.batch_run_sources_device(tb, [1000, 2000, 3000], [d.ptr.value for d in dev], [len(s) for s in srcs], tb, [AIR] * 3)
    for b in range(3):
        gpu = engine.batch_result_copy(b, len(srcs[b]))
        if len(srcs[b]) == 0:
            assert gpu.count == 0
            continue
        assert_same(gpu, O.tick(mdl, nd, nd.packets(srcs[b], start_us=tb[b], air_us=AIR)), "tick %d" % b)
    for d in dev:
        d.free()


def test_batch_refusals(engine, rsa, O):
    from radio_sim_amd import _lib
    n = 500
    nd = random_nodes(O, n, 200.0, seed=1)
    configure_engine(engine, nd, "logdist", dict(ld_flags=1))
    src = DeviceArray(np.arange(10, dtype=np.int32))
    args = ([0], [1000], [src.ptr.value], [10], [0], [AIR])
    recs = DeviceArray(to_tx_records(rsa, nd.packets(np.arange(10), 0, 320)))
    with pytest.raises(rsa.RadioMediumError) as e:                 # SINR records: overlapping ticks are refused at once
        engine.batch_run_device([0, 500], [1000, 1500], [recs.ptr.value] * 2, [10, 10])
    assert e.value.code == _lib.RM_ERR_STATE
    engine.batch_run_device([0], [200], [recs.ptr.value], [10])    # frames of 320 us in a tick of 200 us: found on the device,
    with pytest.raises(rsa.RadioMediumError) as e:                 # reported when the tick is read
        engine.batch_result_copy(0, 10)
    assert e.value.code == _lib.RM_ERR_STATE
    engine.batch_run_device([0], [320], [recs.ptr.value], [10])
    assert engine.batch_result_count(0)[1] == 0
    # SINR: tick 0's frames (8128 us) outlive their tick -- taken since ABI version 4 (rm_airbatch.hip; tests/test_gpu_overlap.py) ...
    engine.batch_run_sources_device([0, 1000], [1000, 2000], [src.ptr.value] * 2, [10, 10], [0, 1000], [AIR, AIR])
    mdl = oracle_model(O, "logdist", dict(ld_flags=1))
    first = nd.packets(np.arange(10), 0, AIR)
    assert_same(engine.batch_result_copy(0, 10), O.tick(mdl, nd, first), "overlapping SINR batch, tick 0")
    assert_same(engine.batch_result_copy(1, 10), O.tick(mdl, nd, np.concatenate([first, nd.packets(np.arange(10), 1000, AIR)]), first_new=10),
                "overlapping SINR batch, tick 1")
    with pytest.raises(rsa.RadioMediumError) as e:                 # ... but not out of time order
        engine.batch_run_sources_device([5000, 4000], [6000, 5000], [src.ptr.value] * 2, [10, 10], [5000, 4000], [AIR, AIR])
    assert e.value.code == _lib.RM_ERR_STATE
    configure_engine(engine, nd, "udgm", dict(udgm_success_ratio_rx=0.5))
    engine.set_partition(0, n // 2)
    with pytest.raises(rsa.RadioMediumError) as e:
        engine.batch_run_sources_device(*args)
    assert e.value.code == _lib.RM_ERR_STATE                       # partition + draws: finish_draws per tick
    engine.set_partition(0, n)
    with pytest.raises(rsa.RadioMediumError) as e:
        k = rsa.MAX_BATCH + 1
        engine.batch_run_sources_device([0] * k, [0] * k, [src.ptr.value] * k, [10] * k, [0] * k, [AIR] * k)
    assert e.value.code == _lib.RM_ERR_INVALID                     # more than RM_MAX_BATCH ticks
    with pytest.raises(rsa.RadioMediumError):
        engine.batch_result_copy(5, 10)                            # no such slot yet
    engine.batch_run_sources_device(*args)
    assert engine.batch_result_copy(0, 10).count > 0
    src.free()
    recs.free()


@pytest.mark.parametrize("acc", ["1", "0"])
def test_sinr_batch_of_self_contained_ticks(engine, rsa, O, monkeypatch, acc):
    """The SINR extension in a batch: allowed when no frame outlives its tick (air time <= tick length),
    every tick then equals the oracle's answer for its own frames; the last tick's frames stay on the
    air for the one-tick-at-a-time calls that follow, and for a batch that begins while they are (the overlap form).
    Both forms of the interference sums: per receiver in Q80 by the exact stage (round 5, the default for ticks named by
    source indices), and through the per-receiver lists (RM_SINR_ACC=0)."""
    from radio_sim_amd import _lib
    monkeypatch.setenv("RM_SINR_ACC", acc)
    n = 6000
    nd = _layout(O, n, seed=41)
    rng = np.random.default_rng(12)
    nd.channel[:] = 11 + rng.integers(0, 3, n)
    params = dict(ld_flags=1, ld_sigma_db=4.0, ld_seed=17, ld_capture_db=3.0)
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    n_ticks = 7
    srcs = _ticks(n, n_ticks, 260, seed=5, ragged=True)
    airs = [960, 320, 1000, 640, 960, 32, 8128]                    # the last one outlives the batch
    dev = [DeviceArray(s) for s in srcs]
    tb = np.arange(n_ticks, dtype=np.int64) * 1000
    engine.batch_run_sources_device(tb, tb + 1000, [d.ptr.value for d in dev], [len(s) for s in srcs], tb, airs)
    interfered = 0
    for b in range(n_ticks):
        cpu = O.tick(mdl, nd, nd.packets(srcs[b], int(tb[b]), airs[b]))
        assert_same(engine.batch_result_copy(b, len(srcs[b])), cpu, "sinr batch tick %d" % b)
        interfered += int((cpu.verdict == O.INTERFERED).sum())
    assert interfered > 50
    # the frames of tick 6 (start 6000, air 8128) are still on the air at t = 7000
    onair = nd.packets(srcs[-1], int(tb[-1]), airs[-1])
    more = np.sort(rng.choice(n, 120, replace=False)).astype(np.int32)
    d_more = DeviceArray(more)
    engine.tick_run_sources_device(7000, 8000, d_more.ptr.value, len(more), 7000, 2048)
    cpu = O.tick(mdl, nd, np.concatenate([onair, nd.packets(more, 7000, 2048)]), first_new=len(onair))
    assert_same(engine.result_copy(len(more)), cpu, "one tick after the batch, with the batch's last frames on the air")
    # ... and a batch that begins at 8000 sees them too (the overlap form, rm_airbatch.hip: refused before ABI version 4)
    engine.batch_run_sources_device([8000], [9000], [d_more.ptr.value], [len(more)], [8000], [320])
    still = np.concatenate([onair, nd.packets(more, 7000, 2048)])
    cpu = O.tick(mdl, nd, np.concatenate([still, nd.packets(more, 8000, 320)]), first_new=len(still))
    assert_same(engine.batch_result_copy(0, len(more)), cpu, "a batch that begins with frames of earlier calls on the air")
    engine.batch_run_sources_device([20000], [21000], [d_more.ptr.value], [len(more)], [20000], [320])   # all expired
    assert_same(engine.batch_result_copy(0, len(more)), O.tick(mdl, nd, nd.packets(more, 20000, 320)), "batch after expiry")
    for d in dev + [d_more]:
        d.free()


def test_sinr_batch_with_lossy_links_draws_from_the_shared_generator(engine, rsa, O):
    """SINR capture and java.util.Random draws in the same medium (rx-loss / tx-loss on some nodes): the
    ticks of such a batch take one launch sequence each (the batched SINR kernels carry no pending
    verdicts); verdicts and the generator must follow the oracle from tick to tick."""
    n = 5000
    nd = _layout(O, n, seed=43, lossy=True)
    params = dict(ld_flags=1, ld_sigma_db=4.0, ld_seed=3, ld_capture_db=3.0)
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    engine.seed(2025)
    state = O.lib().orc_jrandom_seed(2025)
    n_ticks = 4
    srcs = _ticks(n, n_ticks, 200, seed=8, ragged=True)
    dev = [DeviceArray(s) for s in srcs]
    tb = np.arange(n_ticks, dtype=np.int64) * 1000
    engine.batch_run_sources_device(tb, tb + 1000, [d.ptr.value for d in dev], [len(s) for s in srcs], tb, [640] * n_ticks)
    drawn = 0
    for b in range(n_ticks):
        cpu = O.tick(mdl, nd, nd.packets(srcs[b], int(tb[b]), 640), rng_state=state)
        drawn += int(cpu.rng_state != state)
        state = cpu.rng_state
        assert_same(engine.batch_result_copy(b, len(srcs[b])), cpu, "lossy sinr batch tick %d" % b)
    assert drawn == n_ticks and engine.rng_state == state
    for d in dev:
        d.free()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("sinr", [False, True])
def test_sharded_batch_equals_global(rsa, O, world, sinr):
    """The multi-GPU batch sequence on one GPU: every 'rank' packs its own transmitters of all ticks with
    rm_pack_tx_batch_device_on, the 'all-gather' + transposition to tick-major order is done on the
    host, every rank sweeps the gathered ticks against its receiver range with rm_batch_run_device;
    the merged links of every tick equal the global oracle run."""
    from radio_sim_amd import dist as D
    from util import KINDS, _PARAM_MAP
    n, n_ticks = 5000, 5
    params = {"ld_sigma_db": 4.0, "ld_seed": 5}
    nd = _layout(O, n, seed=31)
    air = AIR
    if sinr:          # the SINR extension: ticks whose frames end with the tick (BASELINE configs[3] shape), 3 channels
        params.update(ld_flags=1, ld_capture_db=3.0)
        nd.channel[:] = 11 + np.random.default_rng(3).integers(0, 3, n)
        air = 1000
    mdl = oracle_model(O, "logdist", params)
    srcs = _ticks(n, n_ticks, 120, seed=7, ragged=True)
    slots = D.slots_needed(n, world, srcs)
    starts = np.arange(n_ticks, dtype=np.int64) * 1000
    engines = []
    try:
        mine_bytes = []
        for r in range(world):
            lo, hi = D.partition(n, r, world)
            eng = rsa.Engine(0)
            engines.append(eng)
            eng.upload_table(nd)
            eng.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
            eng.set_partition(lo, hi - lo)
            padded = np.stack([D.pad_sources(s[(s >= lo) & (s < hi)], slots) for s in srcs])     # [ticks][slots]
            d_src, d_out = DeviceArray(padded), DeviceArray(nbytes=n_ticks * slots * 64)
            eng.pack_tx_batch_device_on(0, d_src.ptr.value, n_ticks, slots, starts, air, d_out.ptr.value)
            mine_bytes.append(DeviceArray.read(d_out.ptr.value, np.uint8, n_ticks * slots * 64).reshape(n_ticks, slots * 64))
            d_src.free()
            d_out.free()
        tick_major = np.ascontiguousarray(np.stack(mine_bytes).transpose(1, 0, 2))                # [ticks][rank][slots*64]
        d_all = DeviceArray(tick_major)
        per_tick = world * slots
        ptrs = d_all.ptr.value + np.arange(n_ticks, dtype=np.uint64) * np.uint64(per_tick * 64)
        for eng in engines:
            eng.batch_run_device(starts, starts + 1000, ptrs, [per_tick] * n_ticks)
        for b in range(n_ticks):
            shards = []
            for eng in engines:
                res = eng.batch_result_copy(b, per_tick)
                shards.append((res.pkt, res.dst, res.verdict, res.rssi, res.sinr))
            pkt, dst, verdict, rssi, sinr_db = D.merge_shard_links(shards, per_tick)
            gathered = D.records_from_bytes(tick_major[b].reshape(-1))
            valid, slot_idx = D.drop_padding(gathered)
            np.testing.assert_array_equal(valid["src"], srcs[b])                                  # canonical order survives
            ref = O.tick(mdl, nd, nd.packets(srcs[b], start_us=int(starts[b]), air_us=air))
            assert ref.count > 100 and len(pkt) == ref.count
            np.testing.assert_array_equal(pkt, slot_idx[ref.pkt])
            np.testing.assert_array_equal(dst, ref.dst)
            np.testing.assert_array_equal(verdict, ref.verdict)
            np.testing.assert_array_equal(rssi, ref.rssi)
            np.testing.assert_array_equal(sinr_db, ref.sinr)
            if sinr:
                assert (ref.verdict == O.INTERFERED).any()
        d_all.free()
    finally:
        for eng in engines:
            eng.close()


def test_batch_capacity_handling_and_unsorted_media(engine, rsa, O):
    """A link capacity too small for one tick of the batch is reported for that tick only; raising
    the capacity re-allocates the slots; media without geometry (N2N: unsorted table, no batched
    kernels) go through the same API."""
    n = 4000
    nd = _layout(O, n, seed=13)
    configure_engine(engine, nd, "udgm", {})
    mdl = oracle_model(O, "udgm", {})
    srcs = [np.arange(0, 40, dtype=np.int32), np.arange(0, n, 2, dtype=np.int32), np.arange(100, 140, dtype=np.int32)]
    dev = [DeviceArray(s) for s in srcs]
    tb = [0, 1000, 2000]
    args = (tb, [1000, 2000, 3000], [d.ptr.value for d in dev], [len(s) for s in srcs], tb, [AIR] * 3)
    engine.set_link_capacity(1 << 14)            # tick 1 (2000 frames x ~20 links) does not fit, ticks 0 and 2 do
    engine.batch_run_sources_device(*args)
    for b in (0, 2):
        assert_same(engine.batch_result_copy(b, len(srcs[b])), O.tick(mdl, nd, nd.packets(srcs[b], tb[b], AIR)), "tick %d" % b)
    assert engine.batch_result_count(1)[1] == 1                     # dropped flag
    with pytest.raises(rsa.RadioMediumError) as e:
        engine.batch_result_copy(1, len(srcs[1]))
    assert e.value.code == -4
    views, status = engine.batch_result_view(3, raise_on_error=False)     # the other slots stay usable
    assert status == [0, -4, 0]
    for b in (0, 2):
        assert_same(views[b], O.tick(mdl, nd, nd.packets(srcs[b], tb[b], AIR)), "tick %d in the host block" % b)
    engine.set_link_capacity(1 << 18)
    engine.batch_run_sources_device(*args)
    for b in range(3):
        assert_same(engine.batch_result_copy(b, len(srcs[b])), O.tick(mdl, nd, nd.packets(srcs[b], tb[b], AIR)), "tick %d" % b)
    # N2N matrix medium over 300 nodes
    m = 300
    nd2 = random_nodes(O, m, 100.0, seed=2)
    mat = np.random.default_rng(4).uniform(0.0, 1.0, (m, m))
    mat[mat < 0.7] = 0.0
    configure_engine(engine, nd2, "n2n", {}, matrix=mat)
    engine.set_link_capacity(1 << 22)            # dense medium: a candidate shard must hold a whole (slab, tile) block
    engine.seed(9)
    state = O.lib().orc_jrandom_seed(9)
    mdl2 = oracle_model(O, "n2n", {}, matrix=mat)
    s2 = [np.arange(5 * b, 5 * b + 12, dtype=np.int32) for b in range(4)]
    dev2 = [DeviceArray(s) for s in s2]
    tb2 = [0, 1000, 2000, 3000]
    engine.batch_run_sources_device(tb2, [t + 1000 for t in tb2], [d.ptr.value for d in dev2], [12] * 4, tb2, [AIR] * 4)
    for b in range(4):
        cpu = O.tick(mdl2, nd2, nd2.packets(s2[b], tb2[b], AIR), rng_state=state)
        state = cpu.rng_state
        assert_same(engine.batch_result_copy(b, 12), cpu, "n2n tick %d" % b)
        assert_same(engine.batch_result_view(4)[0][b], cpu, "n2n tick %d in the host block" % b)
    assert engine.rng_state == state
    for d in dev + dev2:
        d.free()


def test_trace_replay_through_a_batch(engine, rsa, O, tmp_path):
    """A compact packet trace (radio_sim_amd.trace) cut into ticks and replayed
    through rm_batch_run_device: every tick equals the oracle's answer for those packets, with the
    trace's per-packet rf-power and start times."""
    import os
    from radio_sim_amd import trace as T
    n = 4000
    nd = _layout(O, n, seed=3)
    rng = np.random.default_rng(2)
    rows = np.zeros(900, dtype=T.TRACE_DTYPE)
    rows["time_us"] = np.sort(rng.integers(0, 12000, 900))
    rows["src"], rows["hex_length"] = rng.integers(0, n, 900), 2 * rng.integers(5, 127, 900)
    rows["txpower"], rows["channel"] = rng.uniform(-5.0, 0.0, 900), 26
    path = os.path.join(str(tmp_path), "replay.rmt")
    T.write_trace(path, rows)
    ticks = T.ticks_of(T.read_trace(path), 1000)
    assert len(ticks) == 12
    params = dict(ld_sigma_db=4.0, ld_seed=8)
    configure_engine(engine, nd, "logdist", params)
    mdl = oracle_model(O, "logdist", params)
    recs = [T.records_of(part, nd) for _, part in ticks]
    dev = [DeviceArray(r) for r in recs]
    tb = [t0 for t0, _ in ticks]
    engine.batch_run_device(tb, [t + 1000 for t in tb], [d.ptr.value for d in dev], [len(r) for r in recs])
    for b, r in enumerate(recs):
        pk = np.zeros(len(r), dtype=O.PACKET_DTYPE)
        for f in ("src", "channel", "x", "y", "z", "txpower", "txprob", "start_us", "air_us"):
            pk[f] = r[f]
        assert_same(engine.batch_result_copy(b, len(r)), O.tick(mdl, nd, pk), "trace tick %d" % b)
    for d in dev:
        d.free()
