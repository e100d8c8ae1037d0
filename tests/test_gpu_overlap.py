"""SINR ticks whose frames outlive their tick, swept as BATCHES (rm_airbatch.hip; BASELINE configs[4]: 8128 us frames over
1000 us ticks).  A frame's verdicts are decided in its first tick against the frames of its own and earlier ticks that are
still on the air, so a batch has to give exactly what the same ticks give one at a time -- and what the oracle gives for
every tick with the full on-air list as interferers (E3 / E4 of DESIGN.md section 6: co-channel, time overlap, interference
floor, half duplex, capture threshold; rssi and sinr bit for bit)."""
import numpy as np
import pytest

from util import KINDS, _PARAM_MAP, oracle_model, DeviceArray

pytestmark = pytest.mark.gpu

TICK = 1000


def _engine(rsa, nd, params, cap=None):
    eng = rsa.Engine(0)
    eng.upload_table(nd)
    eng.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
    if cap:
        eng.set_link_capacity(cap)
    return eng


def _same(gpu, cpu, what):
    assert gpu.count == cpu.count, (what, gpu.count, cpu.count)
    np.testing.assert_array_equal(gpu.pkt, cpu.pkt, err_msg=what)
    np.testing.assert_array_equal(gpu.dst, cpu.dst, err_msg=what)
    np.testing.assert_array_equal(gpu.rssi, cpu.rssi, err_msg=what)
    np.testing.assert_array_equal(gpu.sinr, cpu.sinr, err_msg=what)
    np.testing.assert_array_equal(gpu.verdict, cpu.verdict, err_msg=what)
    np.testing.assert_array_equal(gpu.pkt_interference, cpu.pkt_interference, err_msg=what)


class Replay:
    """the oracle's view of the same run: the frames on the air, tick by tick"""

    def __init__(self, O, nd, params):
        self.O, self.nd, self.mdl = O, nd, oracle_model(O, "logdist", params)
        self.onair = np.zeros(0, dtype=O.PACKET_DTYPE)

    def tick(self, t_begin, srcs, start, air, check=True):
        O = self.O
        self.onair = self.onair[self.onair["start_us"] + self.onair["air_us"] > t_begin]
        new = self.nd.packets(np.asarray(srcs, dtype=np.int32), start, air)
        cpu = None
        if check:
            cpu = O.tick_mt(self.mdl, self.nd, np.concatenate([self.onair, new]), first_new=len(self.onair), cap=1 << 22)
        self.onair = np.concatenate([self.onair, new])
        return cpu


def _run_batch(eng, dev, ticks, t0, air):
    """ticks: list of source arrays; tick k begins (and its frames start) at t0 + k * TICK; air: scalar or per tick"""
    n = len(ticks)
    starts = [t0 + k * TICK for k in range(n)]
    airs = [air] * n if np.isscalar(air) else list(air)
    arrs = [DeviceArray(np.asarray(s, dtype=np.int32)) if len(s) else None for s in ticks]
    dev.extend(a for a in arrs if a is not None)
    eng.batch_run_sources_device(starts, [s + TICK for s in starts], [a.ptr.value if a is not None else 0 for a in arrs],
                                 [len(s) for s in ticks], starts, airs)
    return starts, airs


def _nodes(O, n, k=20.0, seed=1, channels=1):
    rng = np.random.default_rng(seed)
    side = 50.0 * np.sqrt(np.pi * n / k)
    nd = O.NodeTable(n)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    if channels > 1:
        nd.channel[:] = 11 + rng.integers(0, channels, n)
    return nd, rng


def test_overlapping_batches_equal_the_oracle_and_the_lone_ticks(rsa, O):
    """40 k nodes, 200 new frames per tick that stay on the air for 8 more ticks; 26 ticks as batches of 8, 10 and 8 (the on-air
    window is carried from batch to batch) -- every tick against the oracle with the full on-air list, and against a second
    context that runs the same ticks one at a time (the tick by scan)."""
    n, t = 40_000, 200
    nd, rng = _nodes(O, n)
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 21}
    eng, lone = _engine(rsa, nd, params), _engine(rsa, nd, params)
    dev = []
    try:
        rep = Replay(O, nd, params)
        all_ticks = [np.sort(rng.choice(n, t, replace=False)).astype(np.int32) for _ in range(26)]
        k0, interfered = 0, 0
        for nb in (8, 10, 8):
            ticks = all_ticks[k0:k0 + nb]
            starts, airs = _run_batch(eng, dev, ticks, k0 * TICK, 8128)
            for b in range(nb):
                cpu = rep.tick(starts[b], ticks[b], starts[b], 8128)
                gpu = eng.batch_result_copy(b, t)
                assert gpu.count > 5000
                _same(gpu, cpu, "batch tick %d" % (k0 + b))
                d = DeviceArray(ticks[b])
                dev.append(d)
                lone.tick_run_sources_device(starts[b], starts[b] + TICK, d.ptr.value, t, starts[b], 8128)
                _same(lone.result_copy(t), cpu, "lone tick %d" % (k0 + b))
                interfered += int((cpu.verdict == O.INTERFERED).sum())
            k0 += nb
        assert interfered > 1000                      # the overlap does interfere
        assert eng.air_batch_stats() == (3, 26) and lone.air_batch_stats() == (0, 0)
    finally:
        for d in dev:
            d.free()
        eng.close()
        lone.close()


def test_channels_air_times_empty_ticks_and_lone_ticks_share_the_window(rsa, O):
    """16 channels; frames of 300 us to 20 ms (the oldest slot a tick looks at is only a bound: every frame's own times are
    tested); a tick without frames; lone ticks before, between and after the batches see the batches' frames and are seen."""
    n = 20_000
    nd, rng = _nodes(O, n, seed=3, channels=16)
    params = {"ld_flags": 1, "ld_sigma_db": 3.0, "ld_seed": 5}
    eng = _engine(rsa, nd, params)
    dev = []
    try:
        rep = Replay(O, nd, params)
        clock = [0]

        def lone_tick(t, air):
            srcs = np.sort(rng.choice(n, t, replace=False)).astype(np.int32)
            d = DeviceArray(srcs)
            dev.append(d)
            t0 = clock[0]
            eng.tick_run_sources_device(t0, t0 + TICK, d.ptr.value, t, t0, air)
            _same(eng.result_copy(t), rep.tick(t0, srcs, t0, air), "lone tick at %d" % t0)
            clock[0] += TICK

        def batch(sizes, airs):
            ticks = [np.sort(rng.choice(n, s, replace=False)).astype(np.int32) for s in sizes]
            starts, airs = _run_batch(eng, dev, ticks, clock[0], airs)
            for b, s in enumerate(sizes):
                cpu = rep.tick(starts[b], ticks[b], starts[b], airs[b])
                gpu = eng.batch_result_copy(b, s, cap=1 << 20)
                _same(gpu, cpu, "batch tick at %d (%d frames, air %d)" % (starts[b], s, airs[b]))
            clock[0] += len(sizes) * TICK

        lone_tick(500, 8128)
        lone_tick(300, 20000)
        batch([400, 250, 0, 600, 64, 1, 300], [8128, 300, 8128, 2500, 20000, 8128, 999])
        lone_tick(450, 8128)
        batch([500] * 5, 8128)
        batch([130] * 3, [700, 700, 700])           # self-contained ticks -- but frames of earlier calls are on the air
        lone_tick(200, 640)
        assert eng.air_batch_stats() == (3, 15)
    finally:
        for d in dev:
            d.free()
        eng.close()


def test_dense_field_many_links_per_frame_and_long_near_lists(rsa, O):
    """A small, loud field: every frame is heard by thousands of receivers (link chunks of 256 per workgroup) and has hundreds of
    co-channel frames on the air within reach (the near list is worked off in passes); with and without the shadowing table."""
    n = 6000
    rng = np.random.default_rng(9)
    nd = O.NodeTable(n)
    nd.x, nd.y = rng.uniform(0, 500.0, n), rng.uniform(0, 500.0, n)
    nd.txpower[:] = rng.choice([0.0, 6.0, 12.0], n)
    for sigma in (4.0, 0.0):
        params = {"ld_flags": 1, "ld_sigma_db": sigma, "ld_seed": 2}
        eng = _engine(rsa, nd, params, cap=1 << 21)
        dev = []
        try:
            rep = Replay(O, nd, params)
            ticks = [np.sort(rng.choice(n, 90, replace=False)).astype(np.int32) for _ in range(7)]
            starts, airs = _run_batch(eng, dev, ticks, 0, 8128)
            most = 0
            for b in range(7):
                cpu = rep.tick(starts[b], ticks[b], starts[b], 8128)
                _same(eng.batch_result_copy(b, 90, cap=1 << 21), cpu, "dense tick %d (sigma %.0f)" % (b, sigma))
                most = max(most, int(np.bincount(cpu.pkt).max()))
            assert most > 1000, most
        finally:
            for d in dev:
                d.free()
            eng.close()


def test_frames_without_a_bound_and_a_moved_transmitter(rsa, O):
    """Path-loss exponent 0 (no frame has a cut-off: nothing is in the grid, every frame looks at every visible frame), and --
    with the usual exponent -- a node that moves between two batches while its frame is on the air: the frame keeps the
    position it was sent from, its links are evaluated against the table as it is now."""
    n = 1500
    nd, rng = _nodes(O, n, seed=4)
    params = {"ld_flags": 1, "ld_sigma_db": 2.0, "ld_seed": 8, "ld_exponent": 0.0, "ld_pl0_db": 70.0}
    eng = _engine(rsa, nd, params, cap=1 << 21)
    dev = []
    try:
        rep = Replay(O, nd, params)
        ticks = [np.sort(rng.choice(n, 12, replace=False)).astype(np.int32) for _ in range(5)]
        starts, airs = _run_batch(eng, dev, ticks, 0, 3000)
        for b in range(5):
            _same(eng.batch_result_copy(b, 12, cap=1 << 21), rep.tick(starts[b], ticks[b], starts[b], 3000), "unbounded tick %d" % b)
    finally:
        for d in dev:
            d.free()
        eng.close()
    n = 30_000
    nd, rng = _nodes(O, n, seed=6)
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 3}
    eng = _engine(rsa, nd, params)
    dev = []
    try:
        rep = Replay(O, nd, params)
        first = [np.sort(rng.choice(n, 300, replace=False)).astype(np.int32) for _ in range(6)]
        starts, airs = _run_batch(eng, dev, first, 0, 8128)
        for b in range(6):
            _same(eng.batch_result_copy(b, 300), rep.tick(starts[b], first[b], starts[b], 8128), "before the move, tick %d" % b)
        # a transmitter of the last tick moves next to a node that will transmit in the next batch; a receiver moves as well
        j, r = int(first[5][7]), int(rng.integers(n))
        second = [np.sort(rng.choice(n, 300, replace=False)).astype(np.int32) for _ in range(6)]
        near = int(second[0][11])
        for node, (x, y) in ((j, (nd.x[near] + 3.0, nd.y[near])), (r, (nd.x[near] - 2.0, nd.y[near] + 1.0))):
            nd.x[node], nd.y[node] = x, y
            eng.update_node(node, nd.x[node], nd.y[node], nd.z[node], nd.txpower[node], int(nd.channel[node]), 1, 1.0, 1.0)
        starts, airs = _run_batch(eng, dev, second, 6 * TICK, 8128)
        for b in range(6):
            _same(eng.batch_result_copy(b, 300), rep.tick(starts[b], second[b], starts[b], 8128), "after the move, tick %d" % b)
    finally:
        for d in dev:
            d.free()
        eng.close()


def test_a_full_pair_list_defers_frames_and_is_grown(rsa, O, monkeypatch):
    """The list of surviving (link, frame) pairs starts far too small (RM_OV_PAIR_CAP): the frames whose pairs do not fit are
    evaluated in place by the second go of the pair kernel -- the results are the oracle's from the first batch on -- and the
    list is doubled for the batches that follow."""
    monkeypatch.setenv("RM_OV_PAIR_CAP", "4096")
    n, t = 20_000, 200
    nd, rng = _nodes(O, n, seed=12)
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 1}
    eng = _engine(rsa, nd, params)
    dev = []
    try:
        rep = Replay(O, nd, params)
        for rnd in range(4):
            ticks = [np.sort(rng.choice(n, t, replace=False)).astype(np.int32) for _ in range(6)]
            starts, airs = _run_batch(eng, dev, ticks, rnd * 6 * TICK, 8128)
            pairs, frames, interferers = eng.air_batch_pairs()
            assert pairs <= 4096 << rnd and frames >= 6 * t   # (what the list held: far fewer than the pairs there are)
            for b in range(6):
                _same(eng.batch_result_copy(b, t), rep.tick(starts[b], ticks[b], starts[b], 8128), "round %d, tick %d" % (rnd, b))
    finally:
        for d in dev:
            d.free()
        eng.close()


@pytest.mark.parametrize("near_lists", [None, "2"])
def test_overlapping_batch_receiver_sharded(rsa, O, monkeypatch, near_lists):
    """Two and three receiver regions on one GPU: every rank sweeps the gathered frames of all ranks ([rank][tick][slot] source
    indices, padding included) against its receivers, keeps the frames on the air that can matter to its region (k_rank_frames:
    at the interference floor, with a margin), and the ranks' links merged by node index are the whole batch's."""
    if near_lists:
        monkeypatch.setenv("RM_NEAR_LISTS", near_lists)   # (the batch filter's near-frame lists at these sizes too: padding frames, regions)
        monkeypatch.setenv("RM_WG_RPT", "4")
    from radio_sim_amd import dist as D
    n, t, nb = 30_000, 240, 7
    nd, rng = _nodes(O, n, seed=14)
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 77}
    rep = Replay(O, nd, params)
    for world in (2, 3):
        rep.onair = rep.onair[:0]
        engs = [_engine(rsa, nd, params) for _ in range(world)]
        dev = []
        try:
            for r, e in enumerate(engs):
                e.set_partition_spatial(r, world)
            own = engs[0].partition_of_nodes(world)
            for rnd in range(2):                      # the second batch begins with the first one's frames on the air
                ticks = [np.sort(rng.choice(n, t, replace=False)).astype(np.int32) for _ in range(nb)]
                slots = max(int((own[s] == r).sum()) for s in ticks for r in range(world)) + 3
                packed = np.full((world, nb, slots), -1, dtype=np.int32)
                order = []
                for b, s in enumerate(ticks):
                    parts = [s[own[s] == r] for r in range(world)]
                    for r in range(world):
                        packed[r, b, :len(parts[r])] = parts[r]
                    order.append(packed[:, b, :].reshape(-1))       # the tick's packet order: rank-major, padding included
                d = DeviceArray(packed.reshape(-1))
                dev.append(d)
                starts = [(rnd * nb + b) * TICK for b in range(nb)]
                for e in engs:
                    e.batch_run_gathered_sources_device(starts, [s + TICK for s in starts], d.ptr.value, world, slots, starts, 8128)
                for b in range(nb):
                    real = order[b] >= 0
                    srcs = order[b][real]
                    cpu = rep.tick(starts[b], srcs, starts[b], 8128)
                    renum = np.cumsum(real) - 1                       # packet number without the padding slots
                    parts = [e.batch_result_copy(b, world * slots) for e in engs]
                    pk = np.concatenate([renum[p.pkt] for p in parts])
                    key = np.lexsort((np.concatenate([p.dst for p in parts]), pk))
                    for f in ("dst", "rssi", "sinr", "verdict"):
                        np.testing.assert_array_equal(np.concatenate([getattr(p, f) for p in parts])[key], getattr(cpu, f),
                                                      err_msg="%d ranks, round %d, tick %d: %s" % (world, rnd, b, f))
                    np.testing.assert_array_equal(pk[key], cpu.pkt)
                    assert sum(p.count for p in parts) == cpu.count > 5000
        finally:
            for d in dev:
                d.free()
            for e in engs:
                e.close()


def test_receivers_that_move_while_selected_frames_are_on_the_air(rsa, O):
    """A partitioned context keeps the frames on the air that can matter to its region as it was when they were transmitted
    (plus RM_RANK_MARGIN, 64 m).  A receiver that moves a few metres in mid-air changes nothing about that -- the batch after
    the move equals the oracle, which evaluates every frame on the air against the node table as it is now.  A receiver that is
    put far outside its rank's region may be reached by frames this rank never kept: every tick of the next batch -- and a lone
    tick -- then reads as RM_ERR_STATE instead of a verdict that silently lacks an interferer; once the frames have left the
    air the rank goes on."""
    import os
    if os.environ.get("RM_RANK_FRAMES") == "0":
        pytest.skip("the run's knobs keep every gathered frame on every rank: nothing is selected, nothing can be missed (tools/knob_sweep.sh)")
    n, t, nb, world = 20_000, 200, 4, 2
    nd, rng = _nodes(O, n, seed=21)
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 5}
    rep = Replay(O, nd, params)
    engs = [_engine(rsa, nd, params) for _ in range(world)]
    dev = []
    try:
        for r, e in enumerate(engs):
            e.set_partition_spatial(r, world)
        own = engs[0].partition_of_nodes(world)

        def batch(k0, check=True, expect_error_on=()):
            ticks = [np.sort(rng.choice(n, t, replace=False)).astype(np.int32) for _ in range(nb)]
            slots = max(int((own[s] == r).sum()) for s in ticks for r in range(world)) + 1
            packed = np.full((world, nb, slots), -1, dtype=np.int32)
            for b, s in enumerate(ticks):
                for r in range(world):
                    mine = s[own[s] == r]
                    packed[r, b, :len(mine)] = mine
            d = DeviceArray(packed.reshape(-1))
            dev.append(d)
            starts = [(k0 + b) * TICK for b in range(nb)]
            for e in engs:
                e.batch_run_gathered_sources_device(starts, [s + TICK for s in starts], d.ptr.value, world, slots, starts, 8128)
            for b in range(nb):
                order = packed[:, b, :].reshape(-1)
                real = order >= 0
                cpu = rep.tick(starts[b], order[real], starts[b], 8128, check=check)
                parts = []
                for r, e in enumerate(engs):
                    if r in expect_error_on:
                        with pytest.raises(rsa.RadioMediumError) as err:
                            e.batch_result_copy(b, world * slots)
                        assert err.value.code == -5 and "left the region" in str(err.value)
                    else:
                        parts.append(e.batch_result_copy(b, world * slots))
                if check and not expect_error_on:
                    renum = np.cumsum(real) - 1
                    pk = np.concatenate([renum[p.pkt] for p in parts])
                    key = np.lexsort((np.concatenate([p.dst for p in parts]), pk))
                    for f in ("dst", "rssi", "sinr", "verdict"):
                        np.testing.assert_array_equal(np.concatenate([getattr(p, f) for p in parts])[key], getattr(cpu, f), err_msg="tick %d %s" % (k0 + b, f))
                    assert cpu.count > 3000

        batch(0)
        # a few metres, in mid-air: every rank hears of it; nothing to complain about
        movers = rng.choice(n, 50, replace=False).astype(np.int32)
        nd.x[movers] += rng.uniform(-5, 5, 50)
        nd.y[movers] += rng.uniform(-5, 5, 50)
        for e in engs:
            e.move_nodes(movers, nd.x[movers], nd.y[movers])
        batch(nb)
        # one of rank 0's receivers is put at the far corner of the other rank's region
        far = int(np.nonzero(own == 0)[0][7])
        other = np.nonzero(own == 1)[0]
        nd.x[far], nd.y[far] = float(nd.x[other].max()), float(nd.y[other].max())
        for e in engs:
            e.move_nodes(np.array([far], dtype=np.int32), nd.x[[far]], nd.y[[far]])
        batch(2 * nb, check=False, expect_error_on=(0,))
        lone = DeviceArray(np.sort(rng.choice(n, 50, replace=False)).astype(np.int32))
        dev.append(lone)
        engs[0].tick_run_sources_device(3 * nb * TICK, (3 * nb + 1) * TICK, lone.ptr.value, 50, 3 * nb * TICK, 320)
        with pytest.raises(rsa.RadioMediumError) as err:
            engs[0].result_copy(50)
        assert err.value.code == -5 and "left the region" in str(err.value)
        # ... and when every frame that was selected before the move has left the air, the rank goes on (its box is what it is now)
        rep.onair = rep.onair[:0]
        batch(40)
    finally:
        for d in dev:
            d.free()
        for e in engs:
            e.close()


@pytest.mark.parametrize("lone_form", ["scan", "lists"])
def test_a_batch_refused_late_leaves_the_window_where_it_was(rsa, O, monkeypatch, lone_form):
    """A batch can be refused after its ticks were planned (here: a tick of more than 8192 frames), with the advice to run the
    ticks one at a time.  The frames of the batch before are then still on the air for the EARLIER ticks of the refused batch:
    the window's clock must not have moved to the batch's last tick, or the first lone tick would retire them (clock went back)
    and their interference would be lost without a word.  Nor may the planned, never launched ticks have left their slots'
    counters on the other parity: the lone ticks through the sweep kernels (`lists`: RM_SINR_SCAN=0) start from them.
    Every lone tick after the refusal against the oracle."""
    if lone_form == "lists":
        monkeypatch.setenv("RM_SINR_SCAN", "0")
    n = 30_000
    nd, rng = _nodes(O, n, seed=9)
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 5}
    import os
    # (without the link-hash table every pair inside the largest shadowed reach is a candidate: a tick of 8200 frames wants the room)
    eng = _engine(rsa, nd, params, cap=(1 << 26) if os.environ.get("RM_NO_SHADOW_TABLE") else (1 << 24))
    dev = []
    try:
        rep = Replay(O, nd, params)
        first = [np.sort(rng.choice(n, 300, replace=False)).astype(np.int32) for _ in range(3)]
        starts, _ = _run_batch(eng, dev, first, 0, 2500)            # frames end at 2500, 3500, 4500
        for b in range(3):
            _same(eng.batch_result_copy(b, 300), rep.tick(starts[b], first[b], starts[b], 2500), "first batch, tick %d" % b)
        second = [np.sort(rng.choice(n, t, replace=False)).astype(np.int32) for t in (300, 300, 8200)]
        with pytest.raises(rsa.RadioMediumError) as e:
            _run_batch(eng, dev, second, 3000, 2500)
        assert e.value.code == -5 and "one at a time" in str(e.value)
        interfered = 0
        for b, srcs in enumerate(second):
            t0 = 3000 + b * TICK
            d = DeviceArray(srcs)
            dev.append(d)
            eng.tick_run_sources_device(t0, t0 + TICK, d.ptr.value, len(srcs), t0, 2500)
            cpu = rep.tick(t0, srcs, t0, 2500)
            _same(eng.result_copy(len(srcs)), cpu, "lone tick %d after the refusal" % b)
            if b < 2:      # (what the frames of the first batch, still on the air, do to the refused batch's earlier ticks)
                interfered += int((cpu.verdict == O.INTERFERED).sum())
        assert interfered > 20
    finally:
        for d in dev:
            d.free()
        eng.close()


def test_refusals(rsa, O):
    """ticks out of time order and links that draw are refused (RM_ERR_STATE), nothing is left half done"""
    n = 5000
    nd, rng = _nodes(O, n, seed=2)
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 1}
    eng = _engine(rsa, nd, params)
    dev = []
    try:
        ticks = [np.sort(rng.choice(n, 50, replace=False)).astype(np.int32) for _ in range(3)]
        arrs = [DeviceArray(s) for s in ticks]
        dev.extend(arrs)
        with pytest.raises(rsa.RadioMediumError) as e:
            eng.batch_run_sources_device([0, 2000, 1000], [1000, 3000, 2000], [a.ptr.value for a in arrs], [50] * 3, [0, 2000, 1000], [8128] * 3)
        assert e.value.code == -5
        nd.rxprob[::3] = 0.5
        eng.upload_table(nd)
        with pytest.raises(rsa.RadioMediumError) as e:
            eng.batch_run_sources_device([0, 1000, 2000], [1000, 2000, 3000], [a.ptr.value for a in arrs], [50] * 3, [0, 1000, 2000], [8128] * 3)
        assert e.value.code == -5
    finally:
        for d in dev:
            d.free()
        eng.close()
