"""The receiver-sharded tick as the LIBRARY runs it (rm_comm_*, rm_dist_*, rm_group_tick_run_sources_device): packing of
a rank's transmitters, the all-gather of the packed blocks -- RCCL itself, bound inside libradiomedium_hip.so, wherever
the box can offer it: a communicator of ONE rank here (one GPU, and RCCL admits one rank per device); several members
of a group on one device exchange by copies on that device -- and the sweep of the gathered frames, against the oracle.
No torch in these processes: the collective is the product's own."""
import os

import numpy as np
import pytest

from util import DeviceArray, KINDS, _PARAM_MAP, assert_same, oracle_model, random_nodes, to_tx_records

pytestmark = pytest.mark.gpu


def _nodes(O, n, seed):
    return random_nodes(O, n, 50.0 * np.sqrt(np.pi * n / 20.0), seed=seed)


def test_gathered_layout_equals_tick_major_records(rsa, O):
    """rm_batch_run_gathered_device reads tick b's frames out of [rank][tick][slot] where rm_batch_run_device wants them
    tick by tick: the same results, and both equal the oracle."""
    n, world, slots, n_ticks = 5000, 3, 7, 5
    nd = _nodes(O, n, 4)
    rng = np.random.default_rng(4)
    eng = rsa.Engine(0)
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["logdist"], ld_sigma_db=4.0, ld_seed=3)
        gathered = np.zeros((world, n_ticks, slots), dtype=rsa.TX_RECORD_DTYPE)
        gathered["src"] = -1
        for r in range(world):
            for b in range(n_ticks):
                k = int(rng.integers(0, slots + 1))
                gathered[r, b, :k] = to_tx_records(rsa, nd.packets(np.sort(rng.choice(n, k, replace=False)), b * 1000, 8128))
        dev = DeviceArray(gathered)
        t0 = np.arange(n_ticks, dtype=np.int64) * 1000
        eng.batch_run_gathered_device(t0, t0 + 1000, dev.ptr.value, world, slots)
        got = [eng.batch_result_copy(b, world * slots) for b in range(n_ticks)]
        tick_major = np.ascontiguousarray(gathered.transpose(1, 0, 2)).reshape(n_ticks, world * slots)
        dev2 = DeviceArray(tick_major)
        ptrs = dev2.ptr.value + np.arange(n_ticks, dtype=np.uint64) * np.uint64(world * slots * 64)
        eng.batch_run_device(t0, t0 + 1000, ptrs, np.full(n_ticks, world * slots, dtype=np.int32))
        mdl = oracle_model(O, "logdist", {"ld_sigma_db": 4.0, "ld_seed": 3})
        for b in range(n_ticks):
            ref = eng.batch_result_copy(b, world * slots)
            assert_same(got[b], ref, "tick %d gathered vs tick-major" % b)
            recs = tick_major[b]
            valid = np.nonzero(recs["src"] >= 0)[0]
            pk = np.zeros(len(valid), dtype=O.PACKET_DTYPE)
            for f in ("src", "channel", "x", "y", "z", "txpower", "txprob", "start_us", "air_us"):
                pk[f] = recs[f][valid]
            cpu = O.tick(mdl, nd, pk)
            assert got[b].count == cpu.count
            np.testing.assert_array_equal(got[b].pkt, valid[cpu.pkt])
            np.testing.assert_array_equal(got[b].dst, cpu.dst)
            np.testing.assert_array_equal(got[b].rssi, cpu.rssi)
        dev.free()
        dev2.free()
    finally:
        eng.close()


def test_gathered_sources_build_the_records_every_rank_would_have_sent(rsa, O):
    """rm_batch_run_gathered_sources_device: the all-gather of a sharded batch carries source indices (4 bytes per frame), and
    every rank builds all ranks' records from its own copy of the node table -- the same ticks as from gathered records, and
    as the oracle has them."""
    n, world, slots, n_ticks = 6000, 3, 9, 7
    nd = _nodes(O, n, 5)
    rng = np.random.default_rng(5)
    nd.txpower[:] = rng.uniform(-10, 0, n)
    nd.channel[rng.random(n) < 0.3] = 11
    eng = rsa.Engine(0)
    try:
        eng.upload_table(nd)
        eng.set_model(KINDS["logdist"], ld_sigma_db=4.0, ld_seed=3)
        idx = np.full((world, n_ticks, slots), -1, dtype=np.int32)
        records = np.zeros((world, n_ticks, slots), dtype=rsa.TX_RECORD_DTYPE)
        records["src"] = -1
        t0 = np.arange(n_ticks, dtype=np.int64) * 1000
        for r in range(world):
            for b in range(n_ticks):
                k = int(rng.integers(0, slots + 1))
                idx[r, b, :k] = np.sort(rng.choice(n, k, replace=False))
                records[r, b, :k] = to_tx_records(rsa, nd.packets(idx[r, b, :k], int(t0[b]), 8128))
        dev_idx, dev_rec = DeviceArray(idx), DeviceArray(records)
        eng.batch_run_gathered_sources_device(t0, t0 + 1000, dev_idx.ptr.value, world, slots, t0, 8128)
        got = [eng.batch_result_copy(b, world * slots) for b in range(n_ticks)]
        eng.batch_run_gathered_device(t0, t0 + 1000, dev_rec.ptr.value, world, slots)
        mdl = oracle_model(O, "logdist", {"ld_sigma_db": 4.0, "ld_seed": 3})
        links = 0
        for b in range(n_ticks):
            assert_same(got[b], eng.batch_result_copy(b, world * slots), "tick %d: from indices vs from records" % b)
            order = idx[:, b, :].reshape(-1)
            valid = np.nonzero(order >= 0)[0]
            cpu = O.tick(mdl, nd, nd.packets(order[valid], int(t0[b]), 8128))
            assert got[b].count == cpu.count
            np.testing.assert_array_equal(got[b].pkt, valid[cpu.pkt])
            np.testing.assert_array_equal(got[b].dst, cpu.dst)
            np.testing.assert_array_equal(got[b].rssi, cpu.rssi)
            links += cpu.count
        assert links > 500
        dev_idx.free()
        dev_rec.free()
    finally:
        eng.close()


def test_dist_calls_through_rccl_with_one_rank(rsa, O):
    """rm_comm_init_rank with a world of one: ncclGetUniqueId, ncclCommInitRank and ncclAllGather really run (RCCL is bound
    by dlopen inside the library); the batch and the single tick -- with java.util.Random draws -- equal the plain calls."""
    assert rsa.Engine.comm_available(), "RCCL could not be bound"
    n, slots, n_ticks = 6000, 40, 6
    nd = _nodes(O, n, 8)
    rng = np.random.default_rng(8)
    nd.rxprob[rng.choice(n, n // 3, replace=False)] = 0.6
    eng, plain = rsa.Engine(0), rsa.Engine(0)
    try:
        for e in (eng, plain):
            e.upload_table(nd)
            e.set_model(KINDS["udgm"], udgm_success_ratio_rx=0.9)
            e.seed(5)
        eng.comm_init_rank(rsa.Engine.comm_unique_id(), 1, 0)
        src = np.full((n_ticks, slots), -1, dtype=np.int32)
        for b in range(n_ticks):
            k = int(rng.integers(slots // 2, slots + 1))
            src[b, :k] = np.sort(rng.choice(n, k, replace=False))
        dev = DeviceArray(src)
        t0 = np.arange(n_ticks, dtype=np.int64) * 1000
        eng.dist_batch_run_sources_device(t0, t0 + 1000, dev.ptr.value, slots, t0, 8128)
        ptrs = dev.ptr.value + np.arange(n_ticks, dtype=np.uint64) * np.uint64(slots * 4)
        plain.batch_run_sources_device(t0, t0 + 1000, ptrs, np.full(n_ticks, slots, dtype=np.int32), t0, np.full(n_ticks, 8128))
        mdl = oracle_model(O, "udgm", {"udgm_success_ratio_rx": 0.9})
        state = O.lib().orc_jrandom_seed(5)
        for b in range(n_ticks):
            a, p = eng.batch_result_copy(b, slots), plain.batch_result_copy(b, slots)
            assert_same(a, p, "batch tick %d" % b)
            valid = np.nonzero(src[b] >= 0)[0]
            cpu = O.tick(mdl, nd, nd.packets(src[b][valid], int(t0[b]), 8128), rng_state=state)
            state = cpu.rng_state
            assert a.count == cpu.count and cpu.pkt_draws.sum() > 0
            np.testing.assert_array_equal(a.dst, cpu.dst)
            np.testing.assert_array_equal(a.verdict, cpu.verdict)
        assert eng.rng_state == plain.rng_state == state
        # one tick at a time, the draws finished inside the call
        for b in range(3):
            eng.dist_tick_run_sources_device(9000 + b * 1000, 10000 + b * 1000, dev.ptr.value + b * slots * 4, slots, 9000 + b * 1000, 320)
            a = eng.result_copy(slots)
            valid = np.nonzero(src[b] >= 0)[0]
            cpu = O.tick(mdl, nd, nd.packets(src[b][valid], 9000 + b * 1000, 320), rng_state=state)
            state = cpu.rng_state
            assert a.count == cpu.count
            np.testing.assert_array_equal(a.pkt, valid[cpu.pkt])
            np.testing.assert_array_equal(a.dst, cpu.dst)
            np.testing.assert_array_equal(a.verdict, cpu.verdict)
            assert eng.rng_state == state
        dev.free()
    finally:
        eng.close()
        plain.close()


@pytest.mark.parametrize("members,spatial", [(1, True), (2, True), (3, True), (3, False), (8, True)])
@pytest.mark.parametrize("kind,params", [("udgm", {}), ("udgm", {"udgm_success_ratio_rx": 0.8}),
                                         ("logdist", {"ld_sigma_db": 4.0, "ld_seed": 5}),
                                         ("logdist", {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 6})])
def test_group_device_resident_tick(rsa, O, members, spatial, kind, params):
    """rm_group_tick_run_sources_device: every member packs the frames of its own transmitters, the packed blocks go round
    (one member: through RCCL, ncclCommInitAll + ncclAllGather; several members on this one GPU: copies on the device),
    every member sweeps; merged by node index the heard links are the oracle's, draws and SINR lists over several ticks
    included."""
    from radio_sim_amd import dist as D
    n = 3000
    nd = _nodes(O, n, 21)
    rng = np.random.default_rng(21)
    if params.get("udgm_success_ratio_rx", 1.0) != 1.0:
        nd.rxprob[rng.choice(n, n // 4, replace=False)] = 0.5
        nd.txprob[rng.choice(n, n // 10, replace=False)] = 0.7
    g = rsa.Group([0] * members, spatial=spatial)
    try:
        g.upload_table(nd)
        g.set_model(KINDS[kind], **{_PARAM_MAP[k]: v for k, v in params.items()})
        g.seed(31)
        assert g.uses_rccl() == (members == 1 and not os.environ.get("RM_GROUP_NO_RCCL"))   # (tools/knob_sweep.sh sets the knob)
        own = D.owners(n, members, positions=(nd.x, nd.y, nd.z)) if spatial else D.owners(n, members)
        mdl = oracle_model(O, kind, params)
        state = O.lib().orc_jrandom_seed(31)
        sinr = params.get("ld_flags", 0) == 1
        onair = np.zeros(0, dtype=O.PACKET_DTYPE)
        for k, air in enumerate((8128, 320, 2048, 8128)):
            t0 = k * 1000
            srcs = np.sort(rng.choice(n, 50, replace=False)).astype(np.int32)
            slots = D.slots_needed(n, members, [srcs], own)
            rows = [D.pad_sources(srcs[own[srcs] == r], slots) for r in range(members)]
            devs = [DeviceArray(r) for r in rows]
            g.tick_run_sources_device(t0, t0 + 1000, [d.ptr.value for d in devs], slots, t0, air)
            got = g.result_copy()
            order = np.concatenate(rows)                     # the tick's packets: member after member, slot after slot
            valid = np.nonzero(order >= 0)[0]
            new = nd.packets(order[valid], t0, air)
            if sinr:
                onair = onair[onair["start_us"] + onair["air_us"] > t0]
                cpu = O.tick(mdl, nd, np.concatenate([onair, new]), first_new=len(onair), rng_state=state)
                onair = np.concatenate([onair, new])
            else:
                cpu = O.tick(mdl, nd, new, rng_state=state)
            state = cpu.rng_state
            assert got.count == cpu.count > 100, (k, got.count, cpu.count)
            np.testing.assert_array_equal(got.pkt, valid[cpu.pkt], err_msg="tick %d" % k)
            np.testing.assert_array_equal(got.dst, cpu.dst, err_msg="tick %d" % k)
            np.testing.assert_array_equal(got.verdict, cpu.verdict, err_msg="tick %d" % k)
            np.testing.assert_array_equal(got.rssi, cpu.rssi, err_msg="tick %d" % k)
            if sinr:
                np.testing.assert_array_equal(got.sinr, cpu.sinr, err_msg="tick %d" % k)
            assert g.rng_state == state
            for d in devs:
                d.free()
    finally:
        g.close()
