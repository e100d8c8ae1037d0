"""Parity and size-independent properties at BASELINE.json's full sizes (configs[2]: 100k nodes /
1000 frames per tick; the 1M-node layout of configs[4]), synthetic inputs of SURVEY.md section 8d."""
import numpy as np
import pytest

from util import to_tx_records, KINDS, _PARAM_MAP, oracle_model, assert_same

pytestmark = pytest.mark.gpu


def _setup(rsa, O, n, idx, kind, params):
    from radio_sim_amd import workload as W
    src_nd = W.make_nodes(n, idx)
    nd = O.NodeTable(n)
    nd.x, nd.y = src_nd.x, src_nd.y
    eng = rsa.Engine(0)
    eng.upload_table(nd)
    eng.set_model(KINDS[kind], **{_PARAM_MAP[k]: v for k, v in params.items()})
    return nd, eng, W


def test_c3_full_tick_equals_oracle(rsa, O):
    """100k nodes, 1000 concurrent frames, log-distance + shadowing: one whole tick, bit for bit."""
    params = {"ld_sigma_db": 4.0, "ld_seed": 0xC0FFEE}
    nd, eng, W = _setup(rsa, O, 100_000, 3, "logdist", params)
    try:
        srcs = W.choose_sources(100_000, 1000, 0xC0FFEE03, 0)
        pk = nd.packets(srcs, 0, W.AIR_US)
        gpu = eng.tick(to_tx_records(rsa, pk), cap=1 << 20)
        cpu = O.tick(oracle_model(O, "logdist", params), nd, pk, cap=1 << 20)
        assert cpu.count > 30_000 and gpu.count == cpu.count
        np.testing.assert_array_equal(gpu.pkt, cpu.pkt)
        np.testing.assert_array_equal(gpu.dst, cpu.dst)
        np.testing.assert_array_equal(gpu.verdict, cpu.verdict)
        np.testing.assert_array_equal(gpu.rssi, cpu.rssi)
        # structure: packet-major, receiver ascending inside a packet, offsets consistent
        assert np.all(np.diff(gpu.pkt) >= 0)
        same = np.diff(gpu.pkt) == 0
        assert np.all(np.diff(gpu.dst)[same] > 0)
        assert gpu.pkt_offset[0] == 0 and gpu.pkt_offset[-1] == gpu.count
        np.testing.assert_array_equal(np.diff(gpu.pkt_offset.astype(np.int64)), np.bincount(gpu.pkt, minlength=1000))
        # idempotence / determinism: the unordered staging never leaks into the result
        for _ in range(3):
            again = eng.tick(to_tx_records(rsa, pk), cap=1 << 20)
            assert again.count == gpu.count and np.array_equal(again.dst, gpu.dst) and np.array_equal(again.rssi, gpu.rssi)
    finally:
        eng.close()


def test_c3_udgm_reference_medium_full_tick(rsa, O):
    nd, eng, W = _setup(rsa, O, 100_000, 3, "udgm", {})
    try:
        srcs = W.choose_sources(100_000, 1000, 0xC0FFEE03, 7)
        pk = nd.packets(srcs, 7000, W.AIR_US)
        gpu = eng.tick(to_tx_records(rsa, pk), cap=1 << 20)
        cpu = O.tick(oracle_model(O, "udgm", {}), nd, pk, cap=1 << 20)
        assert cpu.count > 15_000 and gpu.count == cpu.count
        np.testing.assert_array_equal(gpu.pkt, cpu.pkt)
        np.testing.assert_array_equal(gpu.dst, cpu.dst)
        np.testing.assert_array_equal(gpu.verdict, cpu.verdict)
        # reciprocity of the unit disc: i hears j  <=>  j hears i   (equal channels, all enabled)
        a = srcs[:200]
        fwd = eng.tick(to_tx_records(rsa, nd.packets(a)), cap=1 << 20)
        pairs = set(zip(a[fwd.pkt].tolist(), fwd.dst.tolist()))
        back_src = np.unique(fwd.dst)[:1500]
        back = eng.tick(to_tx_records(rsa, nd.packets(back_src)), cap=1 << 20)
        heard_back = set(zip(back_src[back.pkt].tolist(), back.dst.tolist()))
        for (i, j) in list(pairs)[:3000]:
            if j in set(back_src.tolist()):
                assert (j, i) in heard_back
    finally:
        eng.close()


def test_range_monotonicity_and_capacity_at_full_size(rsa, O):
    """heard(range r) is a subset of heard(range r' > r); a too small link capacity is reported,
    never silently truncated."""
    nd, eng, W = _setup(rsa, O, 100_000, 3, "udgm", {})
    try:
        srcs = W.choose_sources(100_000, 300, 1, 1)
        recs = to_tx_records(rsa, nd.packets(srcs))
        small = eng.tick(recs, cap=1 << 20)
        eng.set_model(KINDS["udgm"], udgm_transmission_range=80.0)
        big = eng.tick(recs, cap=1 << 20)
        s = set(zip(small.pkt.tolist(), small.dst.tolist()))
        b = set(zip(big.pkt.tolist(), big.dst.tolist()))
        assert s < b and len(b) > 2 * len(s)
        eng.set_link_capacity(1000)
        with pytest.raises(rsa.RadioMediumError) as e:
            eng.tick(recs, cap=1 << 20)
        assert e.value.code == -4
    finally:
        eng.close()


def test_one_million_nodes_whole_tick_against_oracle(rsa, O):
    """1M-node layout (configs[4] size): a tick of 1000 frames, every packet checked link by link against the (threaded)
    oracle."""
    params = {"ld_sigma_db": 4.0, "ld_seed": 5}
    n = 1_000_000
    nd, eng, W = _setup(rsa, O, n, 5, "logdist", params)
    try:
        srcs = W.choose_sources(n, 1000, 0xC0FFEE05, 0)
        pk = nd.packets(srcs, 0, W.AIR_US)
        gpu = eng.tick(to_tx_records(rsa, pk), cap=1 << 20)
        assert gpu.count > 30_000
        assert np.all(np.diff(gpu.pkt) >= 0) and gpu.pkt_offset[-1] == gpu.count
        cpu = O.tick_mt(oracle_model(O, "logdist", params), nd, pk, cap=1 << 20)
        assert cpu.count == gpu.count
        np.testing.assert_array_equal(gpu.pkt, cpu.pkt)
        np.testing.assert_array_equal(gpu.dst, cpu.dst)
        np.testing.assert_array_equal(gpu.verdict, cpu.verdict)
        np.testing.assert_array_equal(gpu.rssi, cpu.rssi)
    finally:
        eng.close()


def _sampled_sinr_check(O, mdl, nd, onair_all, new_all, gpu, sample):
    """Oracle verdicts for the sampled new frames, with every other on-air frame as interferer."""
    others = np.delete(np.arange(len(new_all)), sample)
    active = np.concatenate([onair_all, new_all[others], new_all[sample]])
    cpu = O.tick(mdl, nd, active, first_new=len(active) - len(sample), cap=1 << 20)
    sel = np.isin(gpu.pkt, sample)
    remap = {int(q): i for i, q in enumerate(sample)}
    got_pkt = np.array([remap[int(q)] for q in gpu.pkt[sel]], dtype=np.int32)
    assert cpu.count == sel.sum() > 100
    np.testing.assert_array_equal(got_pkt, cpu.pkt)
    np.testing.assert_array_equal(gpu.dst[sel], cpu.dst)
    np.testing.assert_array_equal(gpu.verdict[sel], cpu.verdict)
    np.testing.assert_array_equal(gpu.rssi[sel], cpu.rssi)
    np.testing.assert_array_equal(gpu.sinr[sel], cpu.sinr)
    return cpu


def _whole_tick_check(O, mdl, nd, onair, new, gpu, what):
    """EVERY new frame of the tick against the oracle (threaded: the tick draws nothing), with every frame on the air as
    a potential interferer."""
    cpu = O.tick_mt(mdl, nd, np.concatenate([onair, new]), first_new=len(onair), cap=1 << 22)
    assert cpu.count == gpu.count > 1000, (what, cpu.count, gpu.count)
    np.testing.assert_array_equal(gpu.pkt, cpu.pkt, err_msg=what)
    np.testing.assert_array_equal(gpu.dst, cpu.dst, err_msg=what)
    np.testing.assert_array_equal(gpu.verdict, cpu.verdict, err_msg=what)
    np.testing.assert_array_equal(gpu.rssi, cpu.rssi, err_msg=what)
    np.testing.assert_array_equal(gpu.sinr, cpu.sinr, err_msg=what)
    np.testing.assert_array_equal(gpu.pkt_interference, cpu.pkt_interference, err_msg=what)
    return cpu


def test_c4_16_channels_sinr_capture_full_size(rsa, O):
    """BASELINE configs[3]: 100k nodes, 5% concurrent Tx (5000 frames), 16 channels, co-channel SINR
    capture -- one full tick on the GPU, ALL 5000 frames checked against the oracle (every link's rssi, sinr and verdict
    bit for bit) with all 5000 frames as potential interferers."""
    from radio_sim_amd import workload as W
    n, t = 100_000, 5000
    src_nd = W.make_nodes(n, 4, channels16=True)
    nd = O.NodeTable(n)
    nd.x, nd.y, nd.channel = src_nd.x, src_nd.y, src_nd.channel
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 0xC0FFEE}
    eng = rsa.Engine(0)
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
        srcs = W.choose_sources(n, t, 0xC0FFEE04, 0)
        pk = nd.packets(srcs, 0, W.AIR_US)
        pk["start_us"] = np.random.default_rng(4).integers(0, 1000, t)
        eng.tick_begin(0, 1000)
        eng.enqueue_records(to_tx_records(rsa, pk))
        gpu = eng.tick_flush(cap=1 << 21)
        assert gpu.count > 10_000          # 5000 frames x ~44 neighbours / 16 channels
        frac_interfered = (gpu.verdict == rsa.INTERFERED).mean()
        assert 0.005 < frac_interfered < 0.6
        cpu = _whole_tick_check(O, oracle_model(O, "logdist", params), nd, pk[:0], pk, gpu, "configs[3], whole tick")
        assert (cpu.verdict == O.INTERFERED).sum() > 100
    finally:
        eng.close()


@pytest.mark.parametrize("acc,lists", [("1", None), ("0", None), ("1", "2")])
def test_c4_batch_of_full_size_ticks(rsa, O, monkeypatch, acc, lists):
    """BASELINE configs[3] as the bench runs it: ticks of 5000 frames over 100k nodes on 16 channels in ONE launch sequence
    (rm_batch_run_sources_device, air time = tick length: self-contained ticks) -- every link of two ticks against the oracle,
    with the interference summed per receiver (the default), through the per-receiver lists (RM_SINR_ACC=0), and with the
    filter's near-frame lists forced."""
    from radio_sim_amd import workload as W
    from util import DeviceArray
    monkeypatch.setenv("RM_SINR_ACC", acc)
    if lists:
        monkeypatch.setenv("RM_NEAR_LISTS", lists)
    n, t, n_ticks = 100_000, 5000, 2
    src_nd = W.make_nodes(n, 4, channels16=True)
    nd = O.NodeTable(n)
    nd.x, nd.y, nd.channel = src_nd.x, src_nd.y, src_nd.channel
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 0xC0FFEE}
    eng = rsa.Engine(0)
    dev = []
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
        eng.set_link_capacity(1 << 22)
        srcs = [W.choose_sources(n, t, 0xC0FFEE04, k) for k in range(n_ticks)]
        dev = [DeviceArray(s) for s in srcs]
        tb = np.arange(n_ticks, dtype=np.int64) * 8128
        eng.batch_run_sources_device(tb, tb + 8128, [d.ptr.value for d in dev], [t] * n_ticks, tb, [8128] * n_ticks)
        mdl = oracle_model(O, "logdist", params)
        for b in range(n_ticks):
            pk = nd.packets(srcs[b], int(tb[b]), 8128)
            gpu = eng.batch_result_copy(b, t, cap=1 << 21)
            assert gpu.count > 10_000
            cpu = _whole_tick_check(O, mdl, nd, pk[:0], pk, gpu, "configs[3], batch tick %d (acc=%s)" % (b, acc))
            assert (cpu.verdict == O.INTERFERED).sum() > 100
    finally:
        for d in dev:
            d.free()
        eng.close()


def test_c5_one_million_nodes_multi_tick_overlap(rsa, O):
    """BASELINE configs[4] shape: 1M nodes, 0.1% new frames per tick (1000), 8128 us frames over
    1000 us ticks: the on-air list grows over the ticks; SINR with time overlap.  Three ticks; the
    third is checked as a whole against the oracle with the full on-air list."""
    from radio_sim_amd import workload as W
    n, t = 1_000_000, 1000
    src_nd = W.make_nodes(n, 5)
    nd = O.NodeTable(n)
    nd.x, nd.y = src_nd.x, src_nd.y
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 11}
    eng = rsa.Engine(0)
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
        eng.set_link_capacity(1 << 23)      # ~1.5 M candidate links per tick with 3000 frames on the air
        rng = np.random.default_rng(5)
        onair = np.zeros(0, dtype=O.PACKET_DTYPE)
        for tick in range(3):
            t0 = tick * 1000
            onair = onair[onair["start_us"] + onair["air_us"] > t0]
            srcs = W.choose_sources(n, t, 0xC0FFEE05, tick)
            new = nd.packets(srcs, 0, W.AIR_US)
            new["start_us"] = t0 + rng.integers(0, 1000, t)
            eng.tick_begin(t0, t0 + 1000)
            eng.enqueue_records(to_tx_records(rsa, new))
            gpu = eng.tick_flush(cap=1 << 20)
            assert gpu.count > 30_000
            if tick == 2:
                assert len(onair) == 2000
                _whole_tick_check(O, oracle_model(O, "logdist", params), nd, onair, new, gpu, "1M nodes, third tick")
            onair = np.concatenate([onair, new])
    finally:
        eng.close()


@pytest.mark.parametrize("form", ["scan", "lists"])
def test_c5_steady_state_nine_thousand_frames_on_the_air(rsa, O, form, monkeypatch):
    """BASELINE configs[4] at its steady state: 8128 us frames over 1000 us ticks keep ~9 ticks of frames (8000-9000)
    on the air; the per-receiver interferer lists live on the device from tick to tick.  WHOLE ticks -- all 1000 new
    frames, every link -- are checked against the oracle with the FULL on-air list (8000+ frames) as interferers: a tick
    of the steady state, the tick in which a node has moved (every list is rebuilt from the frames on the air) and the
    one after it, and a tick after the lists' entry ring has wrapped round (link capacity 2^24: ~0.45 M entries per tick)."""
    from radio_sim_amd import workload as W
    n, t = 1_000_000, 1000
    src_nd = W.make_nodes(n, 5)
    nd = O.NodeTable(n)
    nd.x, nd.y = src_nd.x, src_nd.y
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 11}
    # the default form of these ticks is the scan (rm_airscan.hip: the interferers found among the 9000 frames themselves,
    # nothing kept per receiver); RM_SINR_SCAN=0 runs the same ticks through the lists, whose ring has to wrap as well
    if form == "lists":
        monkeypatch.setenv("RM_SINR_SCAN", "0")
    n_ticks = 60 if form == "lists" else 16
    eng = rsa.Engine(0)
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
        eng.set_link_capacity(1 << 24)      # the lists' ring: 2^24 entries (65536 per sub-ring) -- it goes round during this run
        rng = np.random.default_rng(6)
        onair = np.zeros(0, dtype=O.PACKET_DTYPE)
        mdl = oracle_model(O, "logdist", params)
        checked = {10: "steady state", 13: "the tick a node moved in (lists rebuilt)", 14: "the tick after the rebuild",
                   59: "after the entry ring wrapped"}
        links = 0
        for tick in range(n_ticks):
            t0 = tick * 1000
            if tick == 13:                      # a receiver moves next to a sender: everything its entries were computed from changed
                j = int(rng.integers(n))
                nd.x[j], nd.y[j] = nd.x[(j + 7) % n] + 2.0, nd.y[(j + 7) % n]
                eng.update_node(j, nd.x[j], nd.y[j], nd.z[j], nd.txpower[j], int(nd.channel[j]), 1, 1.0, 1.0)
            onair = onair[onair["start_us"] + onair["air_us"] > t0]
            srcs = W.choose_sources(n, t, 0xC0FFEE05, tick)
            new = nd.packets(srcs, 0, W.AIR_US)
            new["start_us"] = t0 + rng.integers(0, 1000, t)
            eng.tick_begin(t0, t0 + 1000)
            eng.enqueue_records(to_tx_records(rsa, new))
            gpu = eng.tick_flush(cap=1 << 20)
            assert gpu.count > 30_000
            links += gpu.count
            if tick in checked:
                assert len(onair) >= 8000, len(onair)
                cpu = _whole_tick_check(O, mdl, nd, onair, new, gpu, "configs[4], tick %d: %s" % (tick, checked[tick]))
                assert (cpu.verdict == O.INTERFERED).sum() > 0      # the overlap does interfere
            onair = np.concatenate([onair, new])
        inc, reb = eng.air_list_stats()
        if form == "scan":
            assert (inc, reb, eng.air_scan_ticks()) == (0, 0, n_ticks)
        else:
            assert reb == 2 and inc == 58, (inc, reb)     # the first tick and the one with the move
            allocated, held = eng.air_ring_stats()
            assert allocated > held > 0, (allocated, held)   # the entry ring has gone round since the rebuild: old entries were reclaimed
    finally:
        eng.close()


def test_c2_full_tick_equals_oracle(rsa, O):
    """BASELINE configs[1] at its own size: 10 000 nodes, 100 concurrent frames, log-distance path loss (sigma = 0),
    seed 0xC0FFEE02 -- the WHOLE tick bit for bit against the oracle, through the host-buffer tick, the
    device-resident tick and one batch."""
    from radio_sim_amd import workload as W
    from util import DeviceArray
    n, t = 10_000, 100
    src_nd = W.make_nodes(n, 2)
    nd = O.NodeTable(n)
    nd.x, nd.y = src_nd.x, src_nd.y
    eng = rsa.Engine(0)
    devs = []
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["logdist"])
        mdl = oracle_model(O, "logdist", {})
        ticks = [W.choose_sources(n, t, 0xC0FFEE02, k) for k in range(3)]
        want = []
        for k, srcs in enumerate(ticks):
            pk = nd.packets(srcs, k * W.TICK_US, W.AIR_US)
            cpu = O.tick(mdl, nd, pk)
            assert cpu.count > 3000
            want.append(cpu)
            assert_same(eng.tick(to_tx_records(rsa, pk), k * W.TICK_US, (k + 1) * W.TICK_US), cpu, "c2 host-buffer tick %d" % k)
            d = DeviceArray(srcs)
            devs.append(d)
            eng.tick_run_sources_device(k * W.TICK_US, (k + 1) * W.TICK_US, d.ptr.value, t, k * W.TICK_US, W.AIR_US)
            assert_same(eng.result_copy(t), cpu, "c2 device-resident tick %d" % k)
        starts = [k * W.TICK_US for k in range(3)]
        eng.batch_run_sources_device(starts, [s + W.TICK_US for s in starts], [d.ptr.value for d in devs], [t] * 3, starts,
                                     [W.AIR_US] * 3)
        for k in range(3):
            assert_same(eng.batch_result_copy(k, t), want[k], "c2 batch slot %d" % k)
    finally:
        for d in devs:
            d.free()
        eng.close()


def test_more_frames_than_the_fused_scans_hold(rsa, O):
    """> 8192 frames in one tick: segment / packet offsets come from the stand-alone scan kernel."""
    n, t = 30_000, 9_000
    rng = np.random.default_rng(12)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.rxprob[:] = np.where(rng.random(n) < 0.7, 1.0, rng.uniform(0, 1, n))
    eng = rsa.Engine(0)
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["udgm"])
        eng.seed(5)
        srcs = np.sort(rng.choice(n, t, replace=False))
        pk = nd.packets(srcs, 0, 8128)
        gpu = eng.tick(to_tx_records(rsa, pk), cap=1 << 21)
        cpu = O.tick(oracle_model(O, "udgm", {}), nd, pk, rng_state=O.lib().orc_jrandom_seed(5), cap=1 << 21)
        assert cpu.count > 150_000 and gpu.count == cpu.count and cpu.pkt_draws.sum() > 10_000
        np.testing.assert_array_equal(gpu.pkt, cpu.pkt)
        np.testing.assert_array_equal(gpu.dst, cpu.dst)
        np.testing.assert_array_equal(gpu.verdict, cpu.verdict)
        assert eng.rng_state == cpu.rng_state
    finally:
        eng.close()


def test_large_grid_path_with_draws_and_partitions(rsa, O):
    """1M receivers (near-list path of the filter) with probabilistic links, whole and in two
    partitions; sampled packets against the oracle, draw order through the generator state."""
    from radio_sim_amd import workload as W
    from radio_sim_amd import dist as D
    from util import DeviceArray
    n, t = 1_000_000, 600
    src_nd = W.make_nodes(n, 5)
    nd = O.NodeTable(n)
    nd.x, nd.y = src_nd.x, src_nd.y
    rng = np.random.default_rng(3)
    nd.rxprob[:] = np.where(rng.random(n) < 0.5, 1.0, 0.5)
    srcs = W.choose_sources(n, t, 77, 0)
    pk = nd.packets(srcs, 0, W.AIR_US)
    recs = to_tx_records(rsa, pk)
    cpu = O.tick(oracle_model(O, "udgm", {}), nd, pk, rng_state=O.lib().orc_jrandom_seed(9), cap=1 << 20)
    assert cpu.pkt_draws.sum() > 3000
    eng = rsa.Engine(0)
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["udgm"])
        eng.seed(9)
        gpu = eng.tick(recs, cap=1 << 20)
        assert gpu.count == cpu.count
        np.testing.assert_array_equal(gpu.dst, cpu.dst)
        np.testing.assert_array_equal(gpu.verdict, cpu.verdict)
        assert eng.rng_state == cpu.rng_state
    finally:
        eng.close()
    engines, shards, counts = [], [], []
    try:
        for r in range(2):
            lo, hi = D.partition(n, r, 2)
            e = rsa.Engine(0)
            e.upload_table(nd)
            e.set_model(KINDS["udgm"])
            e.set_partition(lo, hi - lo)
            e.seed(9)
            e.tick_begin(0, 1000)
            e.enqueue_records(recs)
            e.tick_run()
            ptr, n_new = e.draw_counts_device()
            counts.append(DeviceArray.read(ptr, np.uint32, n_new))
            engines.append(e)
        for r, e in enumerate(engines):
            e.finish_draws(np.stack(counts), 2, r)
            res = e.result_copy(t, cap=1 << 20)
            shards.append((res.pkt, res.dst, res.verdict, res.rssi, res.sinr))
            assert e.rng_state == cpu.rng_state
        merged = D.merge_shard_links(shards, t)
        np.testing.assert_array_equal(merged[1], cpu.dst)
        np.testing.assert_array_equal(merged[2], cpu.verdict)
    finally:
        for e in engines:
            e.close()


def test_c3_full_size_batch_of_64_ticks(rsa, O):
    """The bench's launch shape: 64 ticks of 1000 frames over 100k nodes in one rm_batch_run_sources_device
    call.  Four whole ticks are compared with the oracle bit for bit, the others through their link counts
    against a second, one-tick-at-a-time run of the engine, and every tick's structure is checked
    (packet-major, receivers ascending, offsets consistent)."""
    from util import DeviceArray
    params = {"ld_sigma_db": 4.0, "ld_seed": 0xC0FFEE}
    nd, eng, W = _setup(rsa, O, 100_000, 3, "logdist", params)
    try:
        eng.set_link_capacity(1 << 20)
        n_ticks = 64
        srcs = [W.choose_sources(100_000, 1000, 0xC0FFEE03, 100 + k) for k in range(n_ticks)]
        dev = [DeviceArray(s) for s in srcs]
        t0 = np.arange(n_ticks, dtype=np.int64) * 1000
        eng.batch_run_sources_device(t0, t0 + 1000, [d.ptr.value for d in dev], [1000] * n_ticks, t0, [W.AIR_US] * n_ticks)
        counts = [eng.batch_result_count(b) for b in range(n_ticks)]
        assert all(drop == 0 for _, drop in counts) and all(30_000 < c < 60_000 for c, _ in counts)
        mdl = oracle_model(O, "logdist", params)
        for b in (0, 21, 42, 63):
            gpu = eng.batch_result_copy(b, 1000, cap=1 << 20)
            cpu = O.tick(mdl, nd, nd.packets(srcs[b], int(t0[b]), W.AIR_US), cap=1 << 20)
            assert gpu.count == cpu.count
            np.testing.assert_array_equal(gpu.pkt, cpu.pkt)
            np.testing.assert_array_equal(gpu.dst, cpu.dst)
            np.testing.assert_array_equal(gpu.verdict, cpu.verdict)
            np.testing.assert_array_equal(gpu.rssi, cpu.rssi)
        for b in range(0, n_ticks, 5):
            gpu = eng.batch_result_copy(b, 1000, cap=1 << 20)
            assert np.all(np.diff(gpu.pkt) >= 0)
            assert np.all(np.diff(gpu.dst)[np.diff(gpu.pkt) == 0] > 0)
            assert gpu.pkt_offset[0] == 0 and gpu.pkt_offset[-1] == gpu.count
            np.testing.assert_array_equal(np.diff(gpu.pkt_offset.astype(np.int64)), np.bincount(gpu.pkt, minlength=1000))
        # the same ticks one at a time on a second context: identical link counts tick by tick
        single = rsa.Engine(0)
        try:
            single.upload_table(nd)
            single.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
            single.set_link_capacity(1 << 20)
            for b in range(n_ticks):
                single.tick_run_sources_device(int(t0[b]), int(t0[b]) + 1000, dev[b].ptr.value, 1000, int(t0[b]), W.AIR_US)
                assert single.result_count() == counts[b]
        finally:
            single.close()
        for d in dev:
            d.free()
    finally:
        eng.close()


def test_c5_batches_of_overlapping_ticks_at_one_million_nodes(rsa, O):
    """BASELINE configs[4] as the bench runs it: 1M nodes, 1000 new frames per tick that stay on the air for 8 more ticks, swept
    as BATCHES of 32 ticks (rm_airbatch.hip).  WHOLE ticks against the oracle with the full on-air list: in the steady state
    (9000 frames on the air: 8 ticks of this batch), the first tick of the next batch (its interferers are all frames of the
    batch before), and -- after a receiver and a transmitter on the air have moved between the batches -- a tick in the middle."""
    from radio_sim_amd import workload as W
    from util import DeviceArray
    n, t, nb = 1_000_000, 1000, 32
    src_nd = W.make_nodes(n, 5)
    nd = O.NodeTable(n)
    nd.x, nd.y = src_nd.x, src_nd.y
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 11}
    eng = rsa.Engine(0)
    dev = []
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
        eng.set_link_capacity(1 << 18)          # per result slot: a tick has ~45 k heard links
        mdl = oracle_model(O, "logdist", params)
        onair = np.zeros(0, dtype=O.PACKET_DTYPE)
        checked = {20: "steady state inside a batch", 32: "first tick of the second batch", 45: "after nodes moved between the batches"}
        for rnd in range(2):
            if rnd == 1:
                j = int(onair["src"][-17])       # a transmitter whose frame is on the air, and some receiver
                r = (j + 12345) % n
                for node in (j, r):
                    nd.x[node], nd.y[node] = nd.x[(node + 7) % n] + 2.0, nd.y[(node + 7) % n]
                    eng.update_node(node, nd.x[node], nd.y[node], nd.z[node], nd.txpower[node], int(nd.channel[node]), 1, 1.0, 1.0)
            ticks = [W.choose_sources(n, t, 0xC0FFEE05, rnd * nb + b) for b in range(nb)]
            arrs = [DeviceArray(s) for s in ticks]
            dev.extend(arrs)
            starts = [(rnd * nb + b) * 1000 for b in range(nb)]
            eng.batch_run_sources_device(starts, [s + 1000 for s in starts], [a.ptr.value for a in arrs], [t] * nb, starts, [W.AIR_US] * nb)
            for b in range(nb):
                k = rnd * nb + b
                onair = onair[onair["start_us"] + onair["air_us"] > starts[b]]
                new = nd.packets(ticks[b], starts[b], W.AIR_US)
                if k in checked:
                    assert len(onair) >= 8000, len(onair)
                    gpu = eng.batch_result_copy(b, t, cap=1 << 20)
                    cpu = _whole_tick_check(O, mdl, nd, onair, new, gpu, "configs[4] in batches, tick %d: %s" % (k, checked[k]))
                    assert (cpu.verdict == O.INTERFERED).sum() > 0
                onair = np.concatenate([onair, new])
        assert eng.air_batch_stats() == (2, 2 * nb)
        pairs, frames, interferers = eng.air_batch_pairs()
        assert frames == 8000 + nb * t and pairs >= interferers > 100 * nb * t
    finally:
        for d in dev:
            d.free()
        eng.close()
