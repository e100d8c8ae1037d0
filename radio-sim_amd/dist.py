"""Receiver-sharded multi-GPU tick: one process per GPU, receivers partitioned over the ranks, one
RCCL all-gather of the tick's Tx records over xGMI.

Every rank owns a set of nodes: their receiver state lives in its HBM (sorted, boxed), and it is
the rank that learns about their transmissions (the emulators of those nodes are attached to its
host process).  Two ways to cut the node set:
  * by node index range [lo, hi) (SURVEY.md 8e as written; `partition`): a rank's receivers lie all
    over the area, every rank's tiles look at every frame;
  * by REGION (`Engine.set_partition_spatial`, `owners(..., engine)`): rank r owns the r-th region of
    the k-d split of all positions, so its filter drops the frames far from its region and the
    per-rank work really is 1/world of the tick's -- at the price of a merge by node index, because
    the ranks' node sets interleave.
Per tick each rank packs the frames of ITS transmitters into a fixed number of 64-byte slots
(padding slots carry src = -1), the ranks all-gather the slots, and every rank then sweeps the full
on-air list against its own receivers.  The gathered order [rank][slot] is the tick's packet order;
every rank's heard links of a packet are ascending in node index, and the global list is their
merge by node index (with index ranges: the ranks' runs one after the other).

Only plumbing here (torch.distributed tensors, numpy); the compute is Engine.tick_run_device.
"""
import numpy as np

from ._lib import TX_RECORD_DTYPE

RECORD_BYTES = TX_RECORD_DTYPE.itemsize  # 64


def partition(n, rank, world):
    """Receiver / source range [lo, hi) owned by `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world


def owner_of(n, world, node):
    """Rank owning `node` under partition()."""
    edges = np.array([(n * r) // world for r in range(world + 1)], dtype=np.int64)
    return np.searchsorted(edges, np.asarray(node, dtype=np.int64), side="right") - 1


def owners(n, world, engine=None, positions=None):
    """int32[n]: the rank owning every node -- by index range, or by region, as Engine.set_partition_spatial(rank, world)
    cuts the table on every rank: `engine` = any context holding the node table, or `positions` = (x, y, z) (host only)."""
    if engine is not None:
        return engine.partition_of_nodes(world)
    if positions is not None:
        from . import _lib
        x, y, z = (np.ascontiguousarray(a, dtype=np.float64) for a in positions)
        out = np.empty(max(n, 1), dtype=np.int32)
        _lib.check(_lib.lib().rm_region_split(n, x.ctypes.data, y.ctypes.data, z.ctypes.data, world, out.ctypes.data))
        return out[:n]
    return owner_of(n, world, np.arange(n)).astype(np.int32)


def slots_needed(n, world, source_lists, owner=None):
    """Smallest slot count that fits every rank's share of every tick (owner: owners(); default index ranges)."""
    need = 1
    for s in source_lists:
        if len(s):
            own = owner_of(n, world, s) if owner is None else owner[np.asarray(s)]
            need = max(need, int(np.bincount(own, minlength=world).max()))
    return need


def pad_sources(local_sources, slots):
    out = np.full(slots, -1, dtype=np.int32)
    out[: len(local_sources)] = local_sources
    return out


def pad_records(local_records, slots):
    """Host-side variant of the device packing: fixed slot count, src = -1 marks padding."""
    out = np.zeros(slots, dtype=TX_RECORD_DTYPE)
    out["src"] = -1
    out[: len(local_records)] = local_records
    return out


def all_gather_records(dist, local, world, out=None, group=None):
    """All-gather one rank's slot buffer (torch uint8 tensor of slots*64 bytes).

    NCCL/RCCL: one all_gather_into_tensor on the current stream.  gloo (CPU tests): list form."""
    import torch
    if out is None:
        out = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
    if local.is_cuda and dist.get_backend() != "gloo":
        dist.all_gather_into_tensor(out, local, group=group)
    elif local.is_cuda:
        # rehearsal on a box without RCCL peers (gloo): stage through the host, synchronously
        torch.cuda.current_stream(local.device).synchronize()
        host = local.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        out.copy_(torch.cat(parts).to(local.device))
    else:
        parts = list(out.view(world, local.numel()).unbind(0))
        dist.all_gather(parts, local, group=group)
    return out


def records_from_bytes(buf):
    """uint8 tensor / array -> structured rm_tx_record array (CPU)."""
    a = buf.cpu().numpy() if hasattr(buf, "cpu") else np.asarray(buf)
    return a.view(TX_RECORD_DTYPE)


def drop_padding(records):
    """(valid records, gathered-slot index of each valid record)."""
    idx = np.nonzero(records["src"] >= 0)[0]
    return records[idx], idx


def merge_shard_links(shards, n_slots):
    """Global packet-major heard-link list from the per-rank lists.

    shards: per rank (pkt, dst, verdict, rssi, sinr) with pkt = gathered slot index.  Every rank's links of a packet are
    ascending in node index and no node belongs to two ranks, so the global list -- the reference's visiting order -- is
    the merge of the ranks' lists by (packet, node index): index ranges give runs that follow each other, regions
    interleave."""
    pkt = np.concatenate([s[0] for s in shards])
    dst = np.concatenate([s[1] for s in shards])
    order = np.lexsort((dst, pkt))
    cols = [np.concatenate([s[i] for s in shards])[order] for i in range(len(shards[0]))]
    assert cols[0].max(initial=-1) < n_slots
    assert len(dst) == 0 or np.all((np.diff(cols[0]) > 0) | (np.diff(cols[1]) > 0)), "a receiver reported by two ranks"
    return cols


def exchange_draw_nodes(counts, node_lists):
    """What the second all-gather of a spatial partition's draw exchange delivers: counts [world, n_new] (uint32) and the
    ranks' lists of drawing nodes (packet-major) -> (all_nodes [world, stride] int32, stride)."""
    world = len(node_lists)
    stride = max(1, max(int(c.sum()) for c in counts))
    out = np.zeros((world, stride), dtype=np.int32)
    for r, lst in enumerate(node_lists):
        out[r, : len(lst)] = lst
    return out, stride


class ShardedTick:
    """Device-resident multi-GPU tick driver used by bench.py (one instance per rank).

    Pipeline: packing + the RCCL all-gather of tick k+1 run on a communication stream while earlier
    ticks are swept (the messages are tiny -- `slots` x 64 B per rank -- so the collective is
    latency-bound and hides behind the sweeps).  With several engine contexts (`engines`, each on
    its own stream) that many ticks are swept concurrently, as on one GPU: the benchmarked medium
    carries no state from tick to tick.  Media with draws or an on-air list must use ONE context.
    Works with world == 1 as well (no collective), which is how the choreography is tested on one GPU."""

    def __init__(self, engines, dist, n, rank, world, slots, device, compute_streams, may_draw=True, batch=1, on_air=False,
                 spatial=False):
        import torch
        self.torch = torch
        if not isinstance(engines, (list, tuple)):
            engines, compute_streams = [engines], [compute_streams]
        self.engines, self.streams = list(engines), list(compute_streams)
        self.eng = self.engines[0]
        self.dist, self.n, self.rank, self.world, self.slots = dist, n, rank, world, slots
        self.lo, self.hi = partition(n, rank, world)
        self.spatial = spatial and world > 1
        if world > 1:
            for e in self.engines:
                if spatial:
                    e.set_partition_spatial(rank, world)     # a region of the plane (the k-d split of all positions)
                else:
                    e.set_partition(self.lo, self.hi - self.lo)
        self.comm = torch.cuda.Stream(device=device)
        ring = len(self.engines) + 1
        # batch > 1: a stage covers `batch` ticks -- ONE all-gather of world x batch x slots source indices (xGMI likes few,
        # larger collectives) and the sweep of all of them (run_batch)
        self.batch = batch
        self.mine = [torch.empty(batch * slots * RECORD_BYTES, dtype=torch.uint8, device=device) for _ in range(ring)]
        self.all = [torch.empty(world * batch * slots * RECORD_BYTES, dtype=torch.uint8, device=device) for _ in range(ring)]
        # one process group (RCCL communicator) per context: their collectives are independent
        self.groups = [dist.new_group() for _ in self.engines] if dist is not None else None
        self.ready = [torch.cuda.Event() for _ in range(ring)]  # gathered records of the buffer are complete
        self.done = [torch.cuda.Event() for _ in range(ring)]   # the sweep that read the buffer has finished
        self.used = [False] * ring
        self.seq = 0           # ticks staged so far
        self.staged = None
        self.cnt_mine = self.cnt_all = None
        self.may_draw = may_draw   # False: the caller knows no java.util.Random draw can happen (saves a call per tick)
        # the SINR medium with frames that outlive their tick: every rank keeps the on-air lists of ITS receivers on its
        # device; a tick's gathered records go to rm_tick_run_records_device (ONE context: the ticks are chained)
        self.on_air = on_air
        if on_air and len(self.engines) != 1:
            raise ValueError("frames that stay on the air chain the ticks: one context")

    def stage(self, dev_src_ptr, t_begin, air_us):
        """Enqueue packing + all-gather of the next tick on the communication stream."""
        torch = self.torch
        k = self.seq
        self.seq += 1
        b = k % len(self.mine)
        eng = self.engines[k % len(self.engines)]
        # the caller keeps the communication stream current (`with torch.cuda.stream(self.comm)` around
        # its tick loop): the collective is enqueued on the current stream, everything else names its stream
        if self.used[b]:
            self.comm.wait_event(self.done[b])         # the sweep that last read this buffer
        eng.pack_tx_device_on(self.comm.cuda_stream, dev_src_ptr, self.slots, t_begin, air_us,
                              self.mine[b].data_ptr())
        if self.dist is not None:
            all_gather_records(self.dist, self.mine[b], self.world, self.all[b])
        else:
            self.all[b].copy_(self.mine[b], non_blocking=True)
        self.ready[b].record(self.comm)
        prev = self.staged
        self.staged = (b, t_begin, k, air_us)
        return prev

    def sweep(self, staged, t_end):
        """Run the sweep of a staged tick on its context's stream."""
        b, t_begin, k, air_us = staged
        eng = self.engines[k % len(self.engines)]
        stream = self.streams[k % len(self.streams)]
        stream.wait_event(self.ready[b])
        if self.on_air:   # (the packed frames all start at t_begin)
            eng.tick_run_records_device(t_begin, t_end, self.all[b].data_ptr(), self.world * self.slots, t_begin + air_us)
        else:
            eng.tick_run_device(t_begin, t_end, self.all[b].data_ptr(), self.world * self.slots)
        if self.may_draw and eng.draws_pending():
            # probabilistic links: the shared java.util.Random is consumed in node order = rank order;
            # one more tiny all-gather (per-packet draw counts), then every rank places its draws
            torch = self.torch
            n_new = self.world * self.slots
            if self.cnt_mine is None or self.cnt_mine.numel() != n_new:
                self.cnt_mine = torch.empty(n_new, dtype=torch.int32, device=self.all[b].device)
                self.cnt_all = torch.empty(self.world * n_new, dtype=torch.int32, device=self.all[b].device)
            with torch.cuda.stream(stream):
                eng.draw_counts_to(self.cnt_mine.data_ptr())
                if self.dist is not None:
                    self.dist.all_gather_into_tensor(self.cnt_all, self.cnt_mine)
                else:
                    self.cnt_all.copy_(self.cnt_mine, non_blocking=True)
                if not self.spatial:
                    eng.finish_draws(self.cnt_all.data_ptr(), self.world, self.rank)
                else:
                    # regions interleave in node order: the drawing links' nodes go round as well, in rows as long as the
                    # longest rank's list (the one host read-back of this path)
                    totals = self.cnt_all.view(self.world, n_new).sum(dim=1)
                    stride = max(1, int(totals.max().item()))
                    mine = torch.zeros(stride, dtype=torch.int32, device=self.all[b].device)
                    k = int(totals[self.rank].item())
                    if k:
                        mine[:k].copy_(self._device_int32(eng.draw_nodes_device(), k))
                    gathered = torch.empty(self.world * stride, dtype=torch.int32, device=mine.device)
                    self.dist.all_gather_into_tensor(gathered, mine)
                    eng.finish_draws_nodes(self.cnt_all.data_ptr(), gathered.data_ptr(), stride, self.world)
        self.done[b].record(stream)
        self.used[b] = True

    def run_batch(self, ctx, dev_src_ptr, t_begins, air_us, tick_us):
        """len(t_begins) <= batch ticks on context `ctx`, everything on that context's own stream:
        ONE all-gather of world x ticks x slots source indices (rows of `slots` int32 at dev_src_ptr) on the
        context's own process group, then the sweep.  No cross-stream events: on this
        runtime an event wait between two streams costs far more than the collective it would hide,
        and with two contexts the other context's sweep runs under this one's all-gather anyway."""
        torch = self.torch
        if self.may_draw and self.world > 1:
            # a partitioned medium whose links draw needs the per-tick draw-count exchange (stage / sweep):
            # rm_batch_run_device would refuse it with RM_ERR_STATE -- say so here, before anything is enqueued
            raise ValueError("run_batch cannot place java.util.Random draws across ranks: use stage() / sweep() per tick "
                             "(or construct ShardedTick with may_draw=False for media without draws)")
        eng, stream = self.engines[ctx], self.streams[ctx]
        nb = len(t_begins)
        if nb > self.batch:
            raise ValueError("run_batch: %d ticks, the buffers hold %d" % (nb, self.batch))
        with torch.cuda.stream(stream):
            # the all-gather carries the ticks' SOURCE INDICES (4 bytes per frame: every rank has the whole node table and
            # builds all ranks' records itself, rm_batch_run_gathered_sources_device) -- a 64-byte record per frame over the
            # links between the GPUs would cost more than the sweep of a rank's share of the receivers
            all_ptr, ranks = dev_src_ptr, 1
            if self.dist is not None:
                mine = self._device_int32(dev_src_ptr, nb * self.slots, stream.device).view(torch.uint8)
                gathered = self.all[ctx][: self.world * nb * self.slots * 4]
                all_gather_records(self.dist, mine, self.world, gathered, group=self.groups[ctx])
                all_ptr, ranks = gathered.data_ptr(), self.world
            eng.batch_run_gathered_sources_device(t_begins, t_begins + tick_us, all_ptr, ranks, self.slots, t_begins, air_us)

    def _device_int32(self, ptr, count, device=None):
        """`count` int32 of device memory somebody else owns as a tensor (no copy)"""
        class _Mem:
            __cuda_array_interface__ = {"shape": (count,), "typestr": "<i4", "data": (ptr, False), "version": 2}
        return self.torch.as_tensor(_Mem(), device=device if device is not None else self.comm.device)

    def run(self, dev_src_ptr, t_begin, t_end, air_us):
        """Unpipelined convenience: stage and sweep one tick."""
        with self.torch.cuda.stream(self.comm):
            self.stage(dev_src_ptr, t_begin, air_us)
            self.sweep(self.staged, t_end)
