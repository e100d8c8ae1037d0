"""World-size-2 gloo test of the multi-GPU exchange path on CPU: packing into slots, the
all-gather of Tx records, canonical order, partition arithmetic and the merge of per-rank link
lists.  The per-rank sweep itself is the GPU engine's job; here the oracle stands in for it
(restricted to the rank's receivers), which is what lets the merged result be checked against the
global oracle run."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, q, mode="index"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as dist
    import radio_sim_amd as rsa
    from radio_sim_amd import dist as D
    from oracle import oracle as O
    from util import to_tx_records

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)                 # same layout on every rank
        nd = O.NodeTable(n)
        side = 50.0 * np.sqrt(np.pi * n / 20.0)
        nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
        lists = [np.sort(rng.choice(n, 30, replace=False)).astype(np.int32) for _ in range(3)]
        lists.append(np.array([n - 1], dtype=np.int32))   # only the last rank transmits
        lists.append(np.zeros(0, dtype=np.int32))         # silent tick
        # who owns which node: index ranges, or regions of the plane (every rank computes the same cut for itself)
        own = D.owners(n, world) if mode == "index" else D.owners(n, world, positions=(nd.x, nd.y, nd.z))
        slots = D.slots_needed(n, world, lists, own)
        lo, hi = D.partition(n, rank, world)
        mdl = O.model(O.MODEL_LOGDIST, ld_sigma_db=4.0, ld_seed=3)
        out = []
        for k, srcs in enumerate(lists):
            mine = srcs[own[srcs] == rank]
            if mode == "index":
                assert np.all(D.owner_of(n, world, mine) == rank) and np.array_equal(mine, srcs[(srcs >= lo) & (srcs < hi)])
            local = D.pad_records(to_tx_records(rsa, nd.packets(mine, k * 1000, 8128)), slots)
            buf = torch.from_numpy(local.view(np.uint8).copy())
            gathered = D.records_from_bytes(D.all_gather_records(dist, buf, world))
            assert len(gathered) == world * slots
            valid, slot_idx = D.drop_padding(gathered)
            # the gathered frames are the tick's packet list: every rank's own frames in node order, rank after rank
            # (index ranges: that is ascending node index overall)
            assert np.array_equal(np.sort(valid["src"]), srcs) and (mode != "index" or np.array_equal(valid["src"], srcs))
            # this rank's sweep: gathered frames against its receivers
            pk = np.zeros(len(valid), dtype=O.PACKET_DTYPE)
            for f in ("src", "channel", "x", "y", "z", "txpower", "txprob", "start_us", "air_us"):
                pk[f] = valid[f]
            r = O.tick(mdl, nd, pk)
            keep = own[r.dst] == rank
            out.append((slot_idx[r.pkt[keep]], r.dst[keep], r.verdict[keep], r.rssi[keep], r.sinr[keep]))
        # the batched exchange of ShardedTick.run_batch: all ticks' slots in ONE all-gather on the
        # context's own process group, then rank-major -> tick-major
        grp = dist.new_group()
        nb, row = len(lists), slots * D.RECORD_BYTES
        mine_all = np.concatenate([D.pad_records(to_tx_records(rsa, nd.packets(s[own[s] == rank], k * 1000, 8128)), slots)
                                   for k, s in enumerate(lists)])
        buf = torch.from_numpy(mine_all.view(np.uint8).copy())
        gathered = D.all_gather_records(dist, buf, world, group=grp)
        tick_major = torch.empty_like(gathered)
        tick_major.view(nb, world, row).copy_(gathered.view(world, nb, row).permute(1, 0, 2))
        for k, srcs in enumerate(lists):
            recs = D.records_from_bytes(tick_major.view(nb, world * row)[k])
            valid, _ = D.drop_padding(recs)
            assert np.array_equal(np.sort(valid["src"]), srcs) and np.all(valid["start_us"] == k * 1000)
        gathered_out = [None] * world
        dist.all_gather_object(gathered_out, out)
        if rank == 0:
            ok = True
            for k, srcs in enumerate(lists):
                merged = D.merge_shard_links([gathered_out[r][k] for r in range(world)], world * slots)
                # the tick's packets in gathered order: rank after rank, each rank's own frames in node order
                order = np.concatenate([srcs[own[srcs] == r] for r in range(world)]).astype(np.int32)
                ref = O.tick(mdl, nd, nd.packets(order, k * 1000, 8128))
                # map gathered slot index back to the position in that list
                slot_of, pos = {}, 0
                for r in range(world):
                    for j in range(int((own[srcs] == r).sum())):
                        slot_of[r * slots + j] = pos
                        pos += 1
                pkt = np.array([slot_of[int(p)] for p in merged[0]], dtype=np.int32)
                if mode != "index" and k == 0:   # regions interleave in node order: the merge is a real permutation
                    ranks = np.concatenate([np.full(len(gathered_out[r][k][0]), r) for r in range(world)])
                    ok &= bool(len(merged[0]) > 100 and not np.array_equal(np.lexsort((ranks, np.concatenate(
                        [gathered_out[r][k][0] for r in range(world)]))), np.lexsort((np.concatenate(
                            [gathered_out[r][k][1] for r in range(world)]), np.concatenate([gathered_out[r][k][0] for r in range(world)])))))
                ok &= (len(pkt) == ref.count and np.array_equal(pkt, ref.pkt) and np.array_equal(merged[1], ref.dst)
                       and np.array_equal(merged[2], ref.verdict) and np.array_equal(merged[3], ref.rssi))
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["index", "spatial"])
def test_world2_exchange_and_merge(mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 700, 11, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True


def test_partition_arithmetic(rsa):
    from radio_sim_amd import dist as D
    for n in (1, 7, 64, 1000, 100000):
        for world in (1, 2, 3, 8):
            edges = [D.partition(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            nodes = np.arange(n)
            own = D.owner_of(n, world, nodes)
            for r, (lo, hi) in enumerate(edges):
                assert np.all(own[lo:hi] == r)
    s = [np.array([0, 1, 2, 900]), np.array([999])]
    assert D.slots_needed(1000, 2, s) == 3
    assert list(D.pad_sources(np.array([4, 5]), 4)) == [4, 5, -1, -1]


def test_region_split_is_balanced_compact_and_deterministic(rsa):
    """rm_region_split (host only): the k-d cut every rank computes for itself.  Parts hold whole groups of 64 nodes in
    proportion, are axis-aligned boxes that do not overlap (the cut is by coordinate, ties by node index), and the answer
    does not depend on anything but the positions."""
    from radio_sim_amd import dist as D
    rng = np.random.default_rng(5)
    for n, world in ((100000, 8), (3001, 3), (700, 2), (50, 4), (1, 2), (0, 3), (1000, 1)):
        x, y, z = rng.uniform(0, 1000, n), rng.uniform(0, 500, n), np.zeros(n)
        own = D.owners(n, world, positions=(x, y, z))
        assert np.array_equal(own, D.owners(n, world, positions=(x.copy(), y.copy(), z.copy())))
        assert len(own) == n and (n == 0 or (own.min() >= 0 and own.max() < world))
        cnt = np.bincount(own, minlength=world)
        groups = -(-n // 64)
        if groups >= world:       # every part gets its share of the groups, within one group
            assert cnt.max() - cnt.min() <= 64 + 63 and cnt.min() > 0, cnt
        if n >= 3001:
            boxes = [(x[own == r].min(), x[own == r].max(), y[own == r].min(), y[own == r].max()) for r in range(world)]
            for a in range(world):
                for b in range(a + 1, world):
                    ox = min(boxes[a][1], boxes[b][1]) - max(boxes[a][0], boxes[b][0])
                    oy = min(boxes[a][3], boxes[b][3]) - max(boxes[a][2], boxes[b][2])
                    assert ox <= 0 or oy <= 0, (a, b, boxes[a], boxes[b])
    # ties: identical coordinates are split by node index
    own = D.owners(256, 2, positions=(np.zeros(256), np.zeros(256), np.zeros(256)))
    assert np.array_equal(own, np.repeat([0, 1], 128))
