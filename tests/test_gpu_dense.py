"""Media in which a frame is heard by a large share of the nodes -- the reference's default NullRadioMedium
(NullRadioMedium.java:47-77: every same-channel node with its radio on hears every frame), a unit disc that covers the
field, a lossless N2N matrix: the tick takes the dense form (rm_dense.hip: node-order evaluation, ordered compaction).
Whole ticks against the oracle, and against the culled paths (RM_DENSE_TICK=0) the same ticks took before."""
import numpy as np
import pytest

from util import KINDS, DeviceArray, assert_same, configure_engine, oracle_model, to_tx_records

pytestmark = pytest.mark.gpu


def _nodes(O, n, side, seed, channels=(26,)):
    rng = np.random.default_rng(seed)
    nd = O.NodeTable(n)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.channel[:] = rng.choice(channels, n)
    nd.enabled[rng.choice(n, n // 40, replace=False)] = 0
    nd.txpower[:] = rng.uniform(-10.0, 0.0, n)
    return nd, rng


@pytest.mark.parametrize("kind,params", [("null", {}), ("udgm", {"udgm_transmission_range": 400.0}), ("udgm_const", {"const_range": 500.0})])
def test_dense_tick_equals_the_oracle(rsa, O, kind, params):
    """20 k nodes, 200 frames, everyone (on the channel, radio on) in range: 4 M links of one tick, every record"""
    n, t = 20_000, 200
    nd, rng = _nodes(O, n, 250.0, 3, channels=(26, 26, 26, 11))
    eng = rsa.Engine(0)
    try:
        configure_engine(eng, nd, kind, params)
        eng.set_link_capacity(1 << 23)
        srcs = np.sort(rng.choice(n, t, replace=False)).astype(np.int32)
        pk = nd.packets(srcs, 0, 8128)
        cpu = O.tick_mt(oracle_model(O, kind, params), nd, pk, cap=1 << 23)
        assert cpu.count > 2_000_000
        gpu = eng.tick(to_tx_records(rsa, pk), cap=1 << 23)
        assert_same(gpu, cpu, kind + " host records")
        np.testing.assert_array_equal(gpu.pkt_offset, np.concatenate([[0], np.cumsum(np.bincount(cpu.pkt, minlength=t))]))
        d = DeviceArray(srcs)
        eng.tick_run_sources_device(0, 1000, d.ptr.value, t, 0, 8128)
        assert_same(eng.result_copy(t, cap=1 << 23), cpu, kind + " source indices on the device")
        d.free()
    finally:
        eng.close()


def test_dense_and_culled_forms_agree_and_alternate(rsa, O, monkeypatch):
    """the same ticks through the dense form and through the forms they took before; ticks of the two kinds one after the
    other on one context (the dense tick leaves the other parity's counters as the sweep's first kernel would)"""
    n = 6000
    nd, rng = _nodes(O, n, 300.0, 5)
    nd.enabled[:] = 1
    eng = rsa.Engine(0)
    try:
        for kind, params in (("udgm", {"udgm_transmission_range": 300.0}), ("null", {})):
            configure_engine(eng, nd, kind, params)
            eng.set_link_capacity(1 << 22)
            mdl = oracle_model(O, kind, params)
            for k in range(6):
                srcs = np.sort(rng.choice(n, 40 + 7 * k, replace=False)).astype(np.int32)
                pk = nd.packets(srcs, k * 1000, 8128)
                cpu = O.tick_mt(mdl, nd, pk, cap=1 << 22)
                monkeypatch.setenv("RM_DENSE_TICK", "1" if k % 2 == 0 else "0")
                assert_same(eng.tick(to_tx_records(rsa, pk), k * 1000, k * 1000 + 1000, cap=1 << 22), cpu, "%s tick %d" % (kind, k))
    finally:
        eng.close()


def test_dense_tick_with_more_cells_than_the_write_pass_sums_itself(rsa, O, monkeypatch):
    """5000 nodes x 1700 frames = 8500 (frame, chunk) cells: beyond 8192 the cells are laid out by the scan kernel instead of
    by the write pass (rm_dense.hip, kDnFusedCells); a short range keeps the tick small, the knob sends it the dense way"""
    n, t = 5000, 1700
    nd, rng = _nodes(O, n, 300.0, 13, channels=(26, 11))
    params = {"udgm_transmission_range": 30.0}
    monkeypatch.setenv("RM_DENSE_TICK", "1")
    eng = rsa.Engine(0)
    try:
        configure_engine(eng, nd, "udgm", params)
        eng.set_link_capacity(1 << 20)
        mdl = oracle_model(O, "udgm", params)
        for k, tt in enumerate((t, 1638, t)): # (1638 * 5 = 8190 cells: the other side of the limit, on the same context)
            srcs = np.sort(rng.choice(n, tt, replace=False)).astype(np.int32)
            pk = nd.packets(srcs, k * 1000, 8128)
            cpu = O.tick_mt(mdl, nd, pk, cap=1 << 20)
            assert cpu.count > 50_000
            gpu = eng.tick(to_tx_records(rsa, pk), k * 1000, k * 1000 + 1000, cap=1 << 20)
            assert_same(gpu, cpu, "tick %d" % k)
            np.testing.assert_array_equal(gpu.pkt_offset, np.concatenate([[0], np.cumsum(np.bincount(cpu.pkt, minlength=tt))]))
    finally:
        eng.close()


def test_dense_tick_on_index_partitions_padding_and_capacity(rsa, O):
    """receivers = an index range (two ranks together = the whole tick), padding records among the frames, and a link capacity
    that is too small is reported, never silently truncated"""
    n, t = 9000, 60
    nd, rng = _nodes(O, n, 200.0, 8)
    params = {"udgm_transmission_range": 400.0}
    mdl = oracle_model(O, "udgm", params)
    srcs = np.sort(rng.choice(n, t, replace=False)).astype(np.int32)
    pk = nd.packets(srcs, 0, 8128)
    recs = to_tx_records(rsa, pk)
    recs["src"][5::11] = -1                       # padding slots of a gathered tick
    real = recs["src"] >= 0
    renum = np.cumsum(real) - 1
    cpu = O.tick_mt(mdl, nd, pk[real], cap=1 << 22)
    parts = []
    for first, count in ((0, 4000), (4000, 5000)):
        eng = rsa.Engine(0)
        try:
            configure_engine(eng, nd, "udgm", params)
            eng.set_partition(first, count)
            d = DeviceArray(recs)
            eng.tick_run_device(0, 1000, d.ptr.value, t)
            parts.append(eng.result_copy(t, cap=1 << 22))
            d.free()
        finally:
            eng.close()
    pk_all = np.concatenate([renum[p.pkt] for p in parts])
    dst_all = np.concatenate([p.dst for p in parts])
    key = np.lexsort((dst_all, pk_all))
    np.testing.assert_array_equal(pk_all[key], cpu.pkt)
    np.testing.assert_array_equal(dst_all[key], cpu.dst)
    np.testing.assert_array_equal(np.concatenate([p.verdict for p in parts])[key], cpu.verdict)
    eng = rsa.Engine(0)
    try:
        configure_engine(eng, nd, "udgm", params)
        eng.set_link_capacity(100_000)
        with pytest.raises(rsa.RadioMediumError) as e:
            eng.tick(recs, cap=1 << 22)
        assert e.value.code == -4
    finally:
        eng.close()


@pytest.mark.parametrize("kind,params", [("null", {}), ("udgm", {"udgm_transmission_range": 400.0})])
def test_the_masks_are_the_result_and_the_records_come_on_request(rsa, O, kind, params, monkeypatch):
    """rm_result_dense (ABI version 5): the dense tick ends with its cells -- per (packet, 1024 consecutive nodes) sixteen lane masks
    and a count; the 17-byte records are written when somebody asks (rm_result_copy here), and are what RM_DENSE_LAZY=0 -- the
    records at once, as before -- gives.  The masks decoded on the host are the oracle's heard sets, packet by packet."""
    import os
    if os.environ.get("RM_DENSE_TICK") == "0":
        pytest.skip("the run's knobs never take the dense form (tools/knob_sweep.sh)")
    n, t = 20_000, 64
    nd, rng = _nodes(O, n, 250.0, 9, channels=(26, 26, 11))
    srcs = np.sort(rng.choice(n, t, replace=False)).astype(np.int32)
    pk = nd.packets(srcs, 0, 8128)
    cpu = O.tick_mt(oracle_model(O, kind, params), nd, pk, cap=1 << 22)
    eng = rsa.Engine(0)
    d = DeviceArray(srcs)
    try:
        configure_engine(eng, nd, kind, params)
        eng.set_link_capacity(1 << 22)
        eng.tick_run_sources_device(0, 1000, d.ptr.value, t, 0, 8128)
        heard, dropped = eng.result_count()            # the totals do not wait for the records
        assert heard == cpu.count and not dropped
        r = eng.result_dense()
        assert r.n_packets == t and r.chunks == -(-n // 1024) and r.rx_first == 0
        masks = DeviceArray.read(r.cell_mask, np.uint64, t * r.chunks * 16).reshape(t, r.chunks * 16)
        counts = DeviceArray.read(r.cell_count, np.uint32, t * r.chunks).reshape(t, r.chunks)
        off = DeviceArray.read(r.pkt_offset, np.uint32, t + 1)
        bits = np.unpackbits(masks.view(np.uint8).reshape(t, -1), axis=1, bitorder="little")[:, :n]
        np.testing.assert_array_equal(bits.sum(axis=1), np.bincount(cpu.pkt, minlength=t))
        np.testing.assert_array_equal(counts.sum(axis=1), np.bincount(cpu.pkt, minlength=t))
        np.testing.assert_array_equal(off, np.concatenate([[0], np.cumsum(np.bincount(cpu.pkt, minlength=t))]))
        q, j = np.nonzero(bits)
        np.testing.assert_array_equal(q, cpu.pkt)
        np.testing.assert_array_equal(j, cpu.dst)
        lazy = eng.result_copy(t, cap=1 << 22)         # ... and now the records
        assert_same(lazy, cpu, kind + " records on request")
        monkeypatch.setenv("RM_DENSE_LAZY", "0")
        eng.tick_run_sources_device(1000, 2000, d.ptr.value, t, 1000, 8128)
        assert_same(eng.result_copy(t, cap=1 << 22), cpu, kind + " records at once")
        monkeypatch.delenv("RM_DENSE_LAZY")
        # a tick that takes another form has no masks to hand out
        configure_engine(eng, nd, "logdist", {})
        eng.tick_run_sources_device(2000, 3000, d.ptr.value, t, 2000, 8128)
        with pytest.raises(rsa.RadioMediumError):
            eng.result_dense()
    finally:
        d.free()
        eng.close()
