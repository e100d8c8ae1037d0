"""Receiver-sharded evaluation on the GPU: several contexts, each with its share of the receivers -- a range of node
indices, or a REGION of the plane (rm_set_partition_spatial) -- see the same gathered Tx slots (padding included); their
heard links, merged by node index, equal the global oracle run.  (The all-gather itself is covered on CPU by
tests/test_dist_gloo.py.)"""
import os

import numpy as np
import pytest

from util import to_tx_records, KINDS, _PARAM_MAP, oracle_model, sinr_lists_forced

pytestmark = pytest.mark.gpu


def owners_and_setter(rsa, D, nd, n, world, mode):
    """(owner of every node, function that puts a context on rank r's share)"""
    if mode == "index":
        own = D.owners(n, world)
        return own, lambda eng, r: eng.set_partition(*(lambda lo, hi: (lo, hi - lo))(*D.partition(n, r, world)))
    probe = rsa.Engine(0)
    try:
        probe.upload_table(nd)
        own = D.owners(n, world, probe)
    finally:
        probe.close()
    return own, lambda eng, r: eng.set_partition_spatial(r, world)


def check_members(eng, own, r, res):
    mine = np.nonzero(own == r)[0]
    np.testing.assert_array_equal(eng.partition_nodes(), mine)       # every rank computes the same cut for itself
    assert res is None or res.count == 0 or np.all(own[res.dst] == r)


@pytest.mark.parametrize("kind,params", [("udgm", {}), ("logdist", {"ld_sigma_db": 4.0, "ld_seed": 5}),
                                         ("logdist", {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 6})])
@pytest.mark.parametrize("world,mode", [(2, "index"), (3, "index"), (2, "spatial"), (3, "spatial"), (8, "spatial")])
def test_sharded_equals_global(rsa, O, kind, params, world, mode):
    from radio_sim_amd import dist as D
    n = 3001
    rng = np.random.default_rng(17)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.channel[:] = 11 + rng.integers(0, 3, n)
    srcs = np.sort(rng.choice(n, 90, replace=False)).astype(np.int32)
    own, put = owners_and_setter(rsa, D, nd, n, world, mode)
    assert np.bincount(own, minlength=world).min() > 0 and (mode == "index" or len(np.unique(np.diff(own))) > 2)
    slots = D.slots_needed(n, world, [srcs], own)
    # what the all-gather would deliver: per rank `slots` records, padded with src = -1
    parts = []
    for r in range(world):
        mine = srcs[own[srcs] == r]
        pk = nd.packets(mine, 0, 8128)
        pk["start_us"] = rng.integers(0, 1000, len(pk))
        parts.append(D.pad_records(to_tx_records(rsa, pk), slots))
    gathered = np.concatenate(parts)
    valid, slot_idx = D.drop_padding(gathered)

    shards = []
    for r in range(world):
        eng = rsa.Engine(0)
        try:
            eng.upload_table(nd)
            eng.set_model(KINDS[kind], **{_PARAM_MAP[k]: v for k, v in params.items()})
            put(eng, r)
            res = eng.tick(gathered)
            check_members(eng, own, r, res)
            shards.append((res.pkt, res.dst, res.verdict, res.rssi, res.sinr))
        finally:
            eng.close()
    pkt, dst, verdict, rssi, sinr = D.merge_shard_links(shards, world * slots)

    pk = np.zeros(len(valid), dtype=O.PACKET_DTYPE)
    for f in ("src", "channel", "x", "y", "z", "txpower", "txprob", "start_us", "air_us"):
        pk[f] = valid[f]
    ref = O.tick(oracle_model(O, kind, params), nd, pk)
    assert ref.count > 300 and len(pkt) == ref.count
    np.testing.assert_array_equal(pkt, slot_idx[ref.pkt])
    np.testing.assert_array_equal(dst, ref.dst)
    np.testing.assert_array_equal(verdict, ref.verdict)
    np.testing.assert_array_equal(rssi, ref.rssi)
    np.testing.assert_array_equal(sinr, ref.sinr)


@pytest.mark.parametrize("kind,params", [("udgm", {"udgm_success_ratio_rx": 0.7}), ("logdist", {"ld_sigma_db": 3.0, "ld_seed": 2}),
                                         ("n2n", {})])
@pytest.mark.parametrize("world,mode", [(2, "index"), (3, "index"), (2, "spatial"), (3, "spatial"), (8, "spatial")])
def test_sharded_probabilistic_links_follow_the_global_draw_order(rsa, O, kind, params, world, mode):
    """Receiver partitions with java.util.Random draws: every rank runs the sweep, the per-packet
    draw counts are exchanged (the all-gather), rm_tick_finish_draws places each rank's draws after
    the lower ranks' -- or, for regions, whose node sets interleave, among the other ranks' by the exchanged
    lists of drawing nodes (rm_tick_finish_draws_nodes) -- verdicts, Tx-failure flags and the generator
    state equal the one-process oracle, on every rank."""
    from radio_sim_amd import dist as D
    from util import DeviceArray
    n = 1500
    rng = np.random.default_rng(5)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.rxprob[:] = np.where(rng.random(n) < 0.4, 1.0, rng.uniform(0, 1, n))
    nd.txprob[:] = np.where(rng.random(n) < 0.5, 1.0, rng.uniform(0, 1.1, n))
    matrix = np.where(rng.random((n, n)) < 0.02, rng.uniform(0, 1.2, (n, n)), 0.0) if kind == "n2n" else None
    ticks = [np.sort(rng.choice(n, 70, replace=False)).astype(np.int32) for _ in range(3)]
    engines = []
    own, put = owners_and_setter(rsa, D, nd, n, world, mode)
    try:
        for r in range(world):
            eng = rsa.Engine(0)
            eng.upload_table(nd)
            eng.set_model(KINDS[kind], **{_PARAM_MAP[k]: v for k, v in params.items()})
            if matrix is not None:
                eng.set_n2n_matrix(matrix)
            put(eng, r)
            eng.seed(77)
            engines.append(eng)
        state = O.lib().orc_jrandom_seed(77)
        mdl = oracle_model(O, kind, params, matrix)
        for k, srcs in enumerate(ticks):
            pk = nd.packets(srcs, 1000 * k, 8128)
            recs = to_tx_records(rsa, pk)
            counts, lists = [], []
            for eng in engines:
                eng.tick_begin(1000 * k, 1000 * k + 1000)
                eng.enqueue_records(recs)
                eng.tick_run()
                assert eng.draws_pending()
                with pytest.raises(rsa.RadioMediumError):      # not final yet
                    eng.result_copy(len(srcs))
                ptr, n_new = eng.draw_counts_device()
                counts.append(DeviceArray.read(ptr, np.uint32, n_new))
                if mode == "spatial":
                    with pytest.raises(rsa.RadioMediumError):  # counts alone cannot place a region's draws
                        eng.finish_draws(np.stack(counts), 1, 0)
                    k_draw = int(counts[-1].sum())
                    lists.append(DeviceArray.read(eng.draw_nodes_device(), np.int32, k_draw) if k_draw else np.zeros(0, np.int32))
            allc = np.stack(counts)                              # what the all-gather delivers
            if mode == "spatial":
                all_nodes, stride = D.exchange_draw_nodes(counts, lists)
            shards = []
            for r, eng in enumerate(engines):
                if mode == "spatial":
                    eng.finish_draws_nodes(allc, all_nodes, stride, world)
                else:
                    eng.finish_draws(allc, world, r)
                res = eng.result_copy(len(srcs))
                shards.append((res.pkt, res.dst, res.verdict, res.rssi, res.sinr))
                last = res
            merged = D.merge_shard_links(shards, len(srcs))
            ref = O.tick(mdl, nd, pk, rng_state=state)
            state = ref.rng_state
            assert ref.pkt_draws.sum() > 100
            np.testing.assert_array_equal(merged[0], ref.pkt)
            np.testing.assert_array_equal(merged[1], ref.dst)
            np.testing.assert_array_equal(merged[2], ref.verdict)
            np.testing.assert_array_equal(last.pkt_interference, ref.pkt_interference)
            for eng in engines:
                assert eng.rng_state == state
    finally:
        for eng in engines:
            eng.close()


def test_pipelined_sharded_driver_on_one_gpu():
    """The two-stream tick driver of the multi-GPU path (pack + gather of tick t+1 overlapping the
    sweep of tick t), run with world == 1: it must see exactly the heard links the plain path sees.
    Run in fresh processes: the driver needs torch, and torch's bundled HIP runtime must be the
    first one loaded in its process."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ([], ["--force-sharded"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c2", "--steps", "40",
                            "--warmup", "5", "--no-cpu-baseline", "--inflight", "1", "--batch", "1"] + extra, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
        outs.append(json.loads(line))
    assert outs[0]["config"]["heard_links_last_tick"] == outs[1]["config"]["heard_links_last_tick"] > 0
    assert outs[1]["value"] > 0


def test_sharded_driver_with_frames_on_the_air_on_one_gpu():
    """configs[4] through the multi-GPU driver with world == 1 (rm_tick_run_records_device under ShardedTick.stage / sweep):
    the same heard links as the plain device path, every tick by scan (or, with the lists: only the first tick builds them)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ([], ["--force-sharded"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c5", "--nodes", "200000", "--steps", "14",
                            "--warmup", "10", "--batch", "1", "--link-capacity", str(1 << 21), "--no-cpu-baseline", "--no-host-transfer"] + extra, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
        outs.append(json.loads(line))
    assert outs[0]["config"]["heard_links_last_tick"] == outs[1]["config"]["heard_links_last_tick"] > 0
    if sinr_lists_forced():                            # (the list form: tools/knob_sweep.sh)
        assert ("1 rebuilt those" if os.environ.get("RM_AIR_LISTS") != "0" else "0 added their frames to per-receiver lists") in outs[1]["config"]["workload"]
    else:
        import re
        m = re.search(r"(\d+) lone ticks among the frames by scan, 0 added their frames to per-receiver lists, 0 rebuilt those", outs[1]["config"]["workload"])
        assert m and int(m.group(1)) >= 24, outs[1]["config"]["workload"]


def test_batched_sharded_driver_through_rccl_with_one_rank():
    """The batched multi-GPU tick driver with a real process group: backend "nccl" (= RCCL) with one
    rank, so that packing, the all-gather on the context's own communicator, the transposition and
    the batched sweep run on one stream exactly as they do with 8 ranks.  stdout must hold the
    result line and nothing else (RCCL prints a banner on its first communicator)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    import socket
    for extra, env in ((["--batch", "16"], {}), (["--force-sharded", "--inflight", "3", "--batch", "16"], {"RM_DIST_SINGLE": "1"})):
        with socket.socket() as sk:                    # (a port of its own per run: the one before may still be in TIME_WAIT)
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c2", "--steps", "8",
                            "--warmup", "2", "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MASTER_PORT=str(port), **env))
        assert p.returncode == 0, p.stderr[-2000:]
        lines = p.stdout.splitlines()
        assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[:500]
        outs.append(json.loads(lines[0]))
    assert outs[0]["config"]["heard_links_last_tick"] == outs[1]["config"]["heard_links_last_tick"] > 0
    assert outs[1]["config"]["ticks_per_launch"] == 16 and outs[1]["value"] > 0


@pytest.mark.parametrize("kind,params", [("udgm", {}), ("logdist", {"ld_sigma_db": 4.0, "ld_seed": 5})])
def test_small_partition_with_many_frames_over_several_ticks(rsa, O, kind, params):
    """A rank that owns fewer than four slabs of receivers (151 here) while every tick carries hundreds
    of frames: the per-frame counters have to be clean again for every tick, also the slots that only
    the slab-less waves of the sweep's first workgroup would clear."""
    n = 8000
    rng = np.random.default_rng(1)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    mdl = oracle_model(O, kind, params)
    for lo, cnt in ((7849, 151), (3000, 70), (10, 200)):
        eng = rsa.Engine(0)
        try:
            eng.upload_table(nd)
            eng.set_model(KINDS[kind], **{_PARAM_MAP[k]: v for k, v in params.items()})
            eng.set_partition(lo, cnt)
            for step, t in enumerate((400, 50, 1, 400, 400)):
                srcs = np.sort(rng.choice(n, t, replace=False)).astype(np.int32)
                pk = nd.packets(srcs, 1000 * step, 320)
                ref = O.tick(mdl, nd, pk)
                keep = (ref.dst >= lo) & (ref.dst < lo + cnt)
                res = eng.tick(to_tx_records(rsa, pk), 1000 * step, 1000 * step + 1000)
                assert res.count == int(keep.sum()), "partition %s tick %d" % ((lo, cnt), step)
                np.testing.assert_array_equal(res.pkt, ref.pkt[keep])
                np.testing.assert_array_equal(res.dst, ref.dst[keep])
                np.testing.assert_array_equal(res.verdict, ref.verdict[keep])
                np.testing.assert_array_equal(res.rssi, ref.rssi[keep])
        finally:
            eng.close()


@pytest.mark.parametrize("world,mode", [(1, "index"), (2, "index"), (3, "index"), (2, "spatial"), (3, "spatial"), (8, "spatial")])
def test_sharded_sinr_with_frames_on_the_air_from_device_records(rsa, O, world, mode):
    """configs[4]'s sharded form: the SINR medium, frames that stay on the air over several ticks, the tick's records
    gathered in device memory (padding included) and handed to every rank's context with rm_tick_run_records_device --
    each context keeps the on-air lists of ITS receivers across the ticks.  Merged, the ranks' links equal the oracle's
    tick over the full on-air list; a receiver moves in mid-run (every context rebuilds its lists from the records it
    kept)."""
    from radio_sim_amd import dist as D
    from util import DeviceArray
    n = 4000
    rng = np.random.default_rng(23)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.channel[:] = 11 + rng.integers(0, 2, n)
    params = {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 9}
    mdl = oracle_model(O, "logdist", params)
    engs = []
    own, put = owners_and_setter(rsa, D, nd, n, world, mode)
    try:
        for r in range(world):
            e = rsa.Engine(0)
            e.upload_table(nd)
            e.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
            if world > 1:
                put(e, r)
            engs.append(e)
        onair = np.zeros(0, dtype=O.PACKET_DTYPE)
        airs = [8128, 2048, 8128, 320, 8128, 4064, 8128, 320, 2048, 8128, 8128, 320]
        interfered = 0
        for tick, air in enumerate(airs):
            t0 = tick * 1000
            if tick == 6:                           # a receiver moves next to a sender: every rank's lists are rebuilt
                j = int(rng.integers(n))
                nd.x[j], nd.y[j] = nd.x[(j + 7) % n] + 2.0, nd.y[(j + 7) % n]
                for e in engs:
                    e.update_node(j, nd.x[j], nd.y[j], nd.z[j], nd.txpower[j], int(nd.channel[j]), 1, 1.0, 1.0)
            srcs = np.sort(rng.choice(n, 60, replace=False)).astype(np.int32)
            slots = D.slots_needed(n, max(world, 1), [srcs], own)
            parts = []
            for r in range(world):
                mine = srcs[own[srcs] == r]
                parts.append(D.pad_records(to_tx_records(rsa, nd.packets(mine, t0, air)), slots))
            gathered = np.concatenate(parts)
            valid, slot_idx = D.drop_padding(gathered)
            dev = DeviceArray(gathered)
            shards = []
            for e in engs:
                e.tick_run_records_device(t0, t0 + 1000, dev.ptr.value, len(gathered), t0 + air)
                res = e.result_copy(len(gathered))
                shards.append((res.pkt, res.dst, res.verdict, res.rssi, res.sinr))
            dev.free()
            pkt, dst, verdict, rssi, sinr = D.merge_shard_links(shards, len(gathered))
            onair = onair[onair["start_us"] + onair["air_us"] > t0]
            new = np.zeros(len(valid), dtype=O.PACKET_DTYPE)
            for f in ("src", "channel", "x", "y", "z", "txpower", "txprob", "start_us", "air_us"):
                new[f] = valid[f]
            ref = O.tick(mdl, nd, np.concatenate([onair, new]), first_new=len(onair))
            onair = np.concatenate([onair, new])
            assert len(pkt) == ref.count > 100, (tick, len(pkt), ref.count)
            np.testing.assert_array_equal(pkt, slot_idx[ref.pkt], err_msg="tick %d" % tick)
            np.testing.assert_array_equal(dst, ref.dst, err_msg="tick %d" % tick)
            np.testing.assert_array_equal(verdict, ref.verdict, err_msg="tick %d" % tick)
            np.testing.assert_array_equal(rssi, ref.rssi, err_msg="tick %d" % tick)
            np.testing.assert_array_equal(sinr, ref.sinr, err_msg="tick %d" % tick)
            interfered += int((ref.verdict == O.INTERFERED).sum())
        assert interfered > 50
        inc, reb = engs[0].air_list_stats()
        if sinr_lists_forced():                        # (the list form, tools/knob_sweep.sh)
            if os.environ.get("RM_AIR_LISTS") != "0":
                assert reb == 2 and inc == len(airs) - 2   # the first tick and the one after the move
        else:                                          # by scan: nothing is kept per receiver, nothing to rebuild
            assert (inc, reb, engs[0].air_scan_ticks()) == (0, 0, len(airs))
    finally:
        for e in engs:
            e.close()


# ---- round 5: a rank's frame list (k_rank_frames) and the node-table digest that rides in the all-gather

def _gathered_batch(rng, n, own, world, n_ticks, t):
    """source indices as the all-gather of a sharded batch leaves them: [rank][tick][slot], -1 = padding"""
    ticks = [np.sort(rng.choice(n, t, replace=False)).astype(np.int32) for _ in range(n_ticks)]
    slots = max(int((own[s] == r).sum()) for s in ticks for r in range(world)) + 2
    packed = np.full((world, n_ticks, slots), -1, dtype=np.int32)
    for b, s in enumerate(ticks):
        for r in range(world):
            mine = s[own[s] == r]
            packed[r, b, :len(mine)] = mine
    return packed, slots


@pytest.mark.parametrize("kind,params,air", [("udgm", {}, 8128), ("logdist", {"ld_sigma_db": 4.0, "ld_seed": 5}, 8128),
                                             ("logdist", {"ld_flags": 1, "ld_sigma_db": 4.0, "ld_seed": 6}, 1000)])
@pytest.mark.parametrize("world,mode", [(3, "spatial"), (8, "spatial"), (3, "index")])
def test_rank_frame_lists_change_nothing(rsa, O, monkeypatch, kind, params, air, world, mode):
    """A partitioned context keeps, of all ranks' gathered frames, only those that can matter to its receivers (k_rank_frames);
    RM_RANK_FRAMES=0 keeps every frame as before.  Every tick's records, packet numbers, offsets by packet and Tx-failure flags
    are the same either way, and the ranks' links merged by node index are the oracle's (16 channels, a dead transmitter and a
    source with txProbability 0 among the frames; self-contained SINR ticks in the third medium)."""
    from radio_sim_amd import dist as D
    from util import DeviceArray
    n, n_ticks, t = 20_000, 5, 300
    rng = np.random.default_rng(23)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.channel[:] = 11 + rng.integers(0, 16, n)
    nd.txprob[rng.choice(n, 40, replace=False)] = 0.0
    nd.enabled[rng.choice(n, 200, replace=False)] = 0
    own, put = owners_and_setter(rsa, D, nd, n, world, mode)
    packed, slots = _gathered_batch(rng, n, own, world, n_ticks, t)
    dev = DeviceArray(packed.reshape(-1))
    t0 = np.arange(n_ticks, dtype=np.int64) * 1000
    mdl = oracle_model(O, kind, params)
    results = {}
    try:
        for knob in ("1", "0"):
            monkeypatch.setenv("RM_RANK_FRAMES", knob)
            per_rank = []
            for r in range(world):
                eng = rsa.Engine(0)
                try:
                    eng.upload_table(nd)
                    eng.set_model(KINDS[kind], **{_PARAM_MAP[k]: v for k, v in params.items()})
                    put(eng, r)
                    eng.batch_run_gathered_sources_device(t0, t0 + 1000, dev.ptr.value, world, slots, t0, air)
                    per_rank.append([eng.batch_result_copy(b, world * slots) for b in range(n_ticks)])
                finally:
                    eng.close()
            results[knob] = per_rank
        heard = 0
        for b in range(n_ticks):
            order = packed[:, b, :].reshape(-1)
            real = np.nonzero(order >= 0)[0]
            cpu = O.tick(mdl, nd, nd.packets(order[real], int(t0[b]), air))
            for r in range(world):
                a, z = results["1"][r][b], results["0"][r][b]
                for f in ("pkt", "dst", "verdict", "rssi", "sinr", "pkt_interference", "pkt_offset"):
                    np.testing.assert_array_equal(getattr(a, f), getattr(z, f), err_msg="rank %d tick %d: %s" % (r, b, f))
                assert a.count == z.count and np.all(own[a.dst] == r)
                np.testing.assert_array_equal(np.diff(a.pkt_offset.astype(np.int64)), np.bincount(a.pkt, minlength=world * slots))
            parts = [results["1"][r][b] for r in range(world)]
            merged = D.merge_shard_links([(p.pkt, p.dst, p.verdict, p.rssi, p.sinr) for p in parts], world * slots)
            np.testing.assert_array_equal(merged[0], real[cpu.pkt])
            for k, f in enumerate(("dst", "verdict", "rssi", "sinr"), start=1):
                np.testing.assert_array_equal(merged[k], getattr(cpu, f), err_msg="tick %d %s" % (b, f))
            np.testing.assert_array_equal(parts[0].pkt_interference[real], cpu.pkt_interference)
            heard += cpu.count
        assert heard > 1000
    finally:
        dev.free()


def test_a_rank_one_node_update_behind_is_found_out(rsa, O):
    """Every rank builds the other ranks' records from ITS copy of the node table (the all-gather carries source indices): a
    rank that missed an rm_node_update would yield silently wrong verdicts.  The ranks' table digests ride in the gathered
    blocks (rm_table_digest, rm_batch_run_gathered_blocks_device): with one context one update behind, every tick of the batch
    reads as RM_ERR_STATE on every rank; after the update has reached it, the same batch is the oracle's."""
    from radio_sim_amd import dist as D
    from util import DeviceArray
    n, world, n_ticks, t = 8000, 3, 4, 120
    rng = np.random.default_rng(31)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    params = {"ld_sigma_db": 4.0, "ld_seed": 9}
    engs = [rsa.Engine(0) for _ in range(world)]
    dev = []
    try:
        for r, e in enumerate(engs):
            e.upload_table(nd)
            e.set_model(KINDS["logdist"], **{_PARAM_MAP[k]: v for k, v in params.items()})
            e.set_partition_spatial(r, world)
        own = engs[0].partition_of_nodes(world)
        assert len({e.table_digest() for e in engs}) == 1
        packed, slots = _gathered_batch(rng, n, own, world, n_ticks, t)
        mover = int(packed[0, 0, 0])                     # a node that transmits in tick 0
        nd.x[mover] += 35.0
        nd.txpower[mover] = -3.0
        for e in engs[:-1]:                              # ... and the last rank's host never hears of its move
            e.update_node(mover, nd.x[mover], nd.y[mover], nd.z[mover], nd.txpower[mover], int(nd.channel[mover]), 1, 1.0, 1.0)
        digests = [e.table_digest() for e in engs]
        assert digests[0] == digests[1] != digests[2]
        t0 = np.arange(n_ticks, dtype=np.int64) * 1000

        def run():
            blocks = rsa.Engine.gather_blocks([packed[r] for r in range(world)], [e.table_digest() for e in engs])
            d = DeviceArray(blocks.reshape(-1))
            dev.append(d)
            for e in engs:
                e.batch_run_gathered_blocks_device(t0, t0 + 1000, d.ptr.value, world, slots, t0, 8128)

        run()
        for e in engs:
            for b in range(n_ticks):
                with pytest.raises(rsa.RadioMediumError) as err:
                    e.batch_result_copy(b, world * slots)
                assert err.value.code == -5 and "node table" in str(err.value)
        engs[-1].update_node(mover, nd.x[mover], nd.y[mover], nd.z[mover], nd.txpower[mover], int(nd.channel[mover]), 1, 1.0, 1.0)
        assert len({e.table_digest() for e in engs}) == 1
        # the digest is a function of the table's content, not of how it came about: a fresh upload of the same table agrees
        fresh = rsa.Engine(0)
        try:
            fresh.upload_table(nd)
            assert fresh.table_digest() == engs[0].table_digest()
        finally:
            fresh.close()
        run()
        mdl = oracle_model(O, "logdist", params)
        for b in range(n_ticks):
            order = packed[:, b, :].reshape(-1)
            real = np.nonzero(order >= 0)[0]
            cpu = O.tick(mdl, nd, nd.packets(order[real], int(t0[b]), 8128))
            parts = [e.batch_result_copy(b, world * slots) for e in engs]
            merged = D.merge_shard_links([(p.pkt, p.dst, p.verdict, p.rssi, p.sinr) for p in parts], world * slots)
            np.testing.assert_array_equal(merged[0], real[cpu.pkt])
            np.testing.assert_array_equal(merged[1], cpu.dst)
            np.testing.assert_array_equal(merged[3], cpu.rssi)
            assert cpu.count > 100
    finally:
        for d in dev:
            d.free()
        for e in engs:
            e.close()
