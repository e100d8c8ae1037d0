"""Object wrapper over one rm_context of libradiomedium_hip.so (no computation here)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ModelParams, DeviceResult, HostResult, TX_RECORD_DTYPE, check


def _ptr(a):
    return None if a is None else a.ctypes.data


_CHAR_BLOCKS = {}


def _wrap(ptr, dtype, count):
    """`count` items of `dtype` at address `ptr`, wrapped in place.  ctypes builds (and keeps) a new array type for
    every distinct length -- tens of microseconds each -- so the wrapper's byte length is rounded up to a power of two
    (numpy only ever looks at the first `count` items)."""
    if count == 0 or not ptr:
        return np.empty(0, dtype=dtype)
    nbytes = count * np.dtype(dtype).itemsize
    size = 1 << max(6, (nbytes - 1).bit_length())
    block = _CHAR_BLOCKS.get(size)
    if block is None:
        block = _CHAR_BLOCKS[size] = C.c_char * size
    return np.frombuffer(block.from_address(ptr), dtype=dtype, count=count)


class TickResult:
    """Heard links of one evaluated tick, packet-major / receiver ascending."""

    def __init__(self, count, pkt, dst, verdict, rssi, sinr, pkt_interference, pkt_offset):
        self.count = count
        self.pkt, self.dst, self.verdict, self.rssi, self.sinr = pkt, dst, verdict, rssi, sinr
        self.pkt_interference, self.pkt_offset = pkt_interference, pkt_offset


class HostTickView:
    """The same fields as TickResult over the engine's pinned host block, each wrapped in place the first time it is read
    (a host loop that only looks at a count or two columns does not pay for seven array wrappers per tick)."""
    _FIELDS = {"pkt": ("pkt", np.int32, 0), "dst": ("dst", np.int32, 0), "verdict": ("verdict", np.uint8, 0),
               "rssi": ("rssi", np.float64, 0), "sinr": ("sinr", np.float64, 0),
               "pkt_interference": ("pkt_interference", np.uint8, 1), "pkt_offset": ("pkt_offset", np.uint32, 2)}

    def __init__(self, r):
        self.count = r.count
        self._n_packets = r.n_packets
        self._ptr = {k: getattr(r, v[0]) for k, v in self._FIELDS.items()}
        self._pkt_rssi = r.pkt_rssi   # (ABI version 5) the reference's media: one rssi per packet, no per-link column

    def __getattr__(self, name):
        spec = HostTickView._FIELDS.get(name)
        if spec is None:
            raise AttributeError(name)
        n = (self.count, self._n_packets, self._n_packets + 1)[spec[2]]
        p = self._ptr[name]
        if name == "pkt" and not p:   # (the block carries no packet column since ABI version 3: it follows from the offsets)
            off = self.pkt_offset.astype(np.int64)
            a = np.repeat(np.arange(self._n_packets, dtype=np.int32), np.diff(off))[:n]
        elif name == "rssi" and not p:   # (one value per packet crossed the link: a link's rssi is its packet's transmit power)
            per_pkt = _wrap(self._pkt_rssi, np.float64, self._n_packets) if self._n_packets else np.zeros(0)
            a = np.repeat(per_pkt, np.diff(self.pkt_offset.astype(np.int64)))[:n]
        else:
            a = np.zeros(n) if (name == "sinr" and not p) else _wrap(p, spec[1], n)
        setattr(self, name, a)
        return a


class Engine:
    """One GPU context == one Simulator's radio medium (RadioMedium.java:35-45)."""

    def __init__(self, device=0):
        self._L = _lib.lib()
        h = C.c_void_p()
        check(self._L.rm_create(device, C.byref(h)))
        self._h = h
        self.n = 0
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._L.rm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- RadioMedium.getName / getBaseRSSI
    def get_name(self):
        return self._L.rm_get_name(self._h).decode()

    def get_base_rssi(self, node=0):
        return self._L.rm_get_base_rssi(self._h, node)

    def set_base_rssi(self, v):
        check(self._L.rm_set_base_rssi(self._h, v))

    # -- model
    @staticmethod
    def default_params(kind):
        p = ModelParams()
        _lib.lib().rm_model_defaults(C.byref(p), kind)
        return p

    def set_model(self, kind, **kw):
        p = self.default_params(kind)
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        check(self._L.rm_set_model(self._h, C.byref(p)))
        return p

    def set_n2n_matrix(self, m):
        m = np.ascontiguousarray(m, dtype=np.float64)
        assert m.ndim == 2 and m.shape[0] == m.shape[1]
        check(self._L.rm_set_n2n_matrix(self._h, m.shape[0], m.ctypes.data))

    # -- java.util.Random
    def seed(self, seed):
        check(self._L.rm_seed(self._h, seed))

    @property
    def rng_state(self):
        s = C.c_uint64()
        check(self._L.rm_get_rng_state(self._h, C.byref(s)))
        return s.value

    @rng_state.setter
    def rng_state(self, v):
        check(self._L.rm_set_rng_state(self._h, v))

    # -- node state
    def upload_nodes(self, x, y, z=None, txpower=None, channel=None, enabled=None, rxprob=None, txprob=None,
                     int_id=None):
        def arr(a, dt):
            return None if a is None else np.ascontiguousarray(a, dtype=dt)
        x = arr(x, np.float64)
        n = len(x)
        args = [x, arr(y, np.float64), arr(z, np.float64), arr(txpower, np.float64), arr(channel, np.int32),
                arr(enabled, np.uint8), arr(rxprob, np.float64), arr(txprob, np.float64), arr(int_id, np.int32)]
        for a in args:
            assert a is None or a.shape == (n,)
        check(self._L.rm_nodes_upload(self._h, n, *[_ptr(a) for a in args]))
        self.n = n

    def upload_table(self, nd):
        """nd: any object with x,y,z,txpower,channel,enabled,rxprob,txprob,int_id arrays."""
        self.upload_nodes(nd.x, nd.y, nd.z, nd.txpower, nd.channel, nd.enabled, nd.rxprob, nd.txprob, nd.int_id)

    def update_node(self, i, x, y, z, txpower, channel, enabled, rxprob, txprob):
        check(self._L.rm_node_update(self._h, i, x, y, z, txpower, channel, enabled, rxprob, txprob))

    def move_nodes(self, nodes, x, y, z=None):
        """New positions of several nodes in one call (rm_nodes_move)."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        z = None if z is None else np.ascontiguousarray(z, dtype=np.float64)
        check(self._L.rm_nodes_move(self._h, len(nodes), _ptr(nodes), _ptr(x), _ptr(y), _ptr(z)))

    def receiver_table_builds(self):
        return int(self._L.rm_receiver_table_builds(self._h))

    def set_partition(self, first, count):
        check(self._L.rm_set_partition(self._h, first, count))

    def set_partition_spatial(self, part, n_parts):
        """this context's receivers = region `part` of `n_parts` of the k-d split over all node positions"""
        check(self._L.rm_set_partition_spatial(self._h, part, n_parts))

    def partition_of_nodes(self, n_parts):
        """part owning every node under set_partition_spatial(., n_parts) (int32[n])"""
        out = np.empty(max(self.n, 1), dtype=np.int32)
        check(self._L.rm_partition_of_nodes(self._h, n_parts, out.ctypes.data))
        return out[: self.n]

    def partition_nodes(self):
        """node indices of this context's receivers, ascending"""
        out = np.empty(max(self.n, 1), dtype=np.int32)
        k = C.c_int32()
        check(self._L.rm_partition_nodes(self._h, out.ctypes.data, len(out), C.byref(k)))
        return out[: k.value].copy()

    def set_link_capacity(self, cap):
        check(self._L.rm_set_link_capacity(self._h, cap))

    def set_time(self, t):
        check(self._L.rm_set_time(self._h, t))

    def set_stream(self, stream_ptr):
        check(self._L.rm_set_stream(self._h, C.c_void_p(stream_ptr)))

    # -- transmit(): one packet
    def transmit(self, src, start_us=0, hex_length=0, txpower=None, channel=None, cap=None):
        cap = cap if cap is not None else max(1, self.n)
        dst = np.empty(cap, dtype=np.int32)
        verdict = np.empty(cap, dtype=np.uint8)
        rssi = np.empty(cap, dtype=np.float64)
        sinr = np.empty(cap, dtype=np.float64)
        cnt = C.c_uint32()
        interf = C.c_uint8()
        tp = C.byref(C.c_double(txpower)) if txpower is not None else None
        ch = C.byref(C.c_int32(channel)) if channel is not None else None
        check(self._L.rm_transmit(self._h, src, start_us, hex_length, tp, ch, dst.ctypes.data, verdict.ctypes.data,
                                  rssi.ctypes.data, sinr.ctypes.data, cap, C.byref(cnt), C.byref(interf)))
        k = cnt.value
        return TickResult(k, np.zeros(k, dtype=np.int32), dst[:k], verdict[:k], rssi[:k], sinr[:k],
                          np.array([interf.value], dtype=np.uint8), np.array([0, k], dtype=np.uint32))

    # -- batched tick (host records)
    def tick_begin(self, t_begin, t_end):
        check(self._L.rm_tick_begin(self._h, t_begin, t_end))
        self._n_new = 0

    def enqueue_tx(self, src, start_us, air_us, txpower=None, channel=None):
        tp = C.byref(C.c_double(txpower)) if txpower is not None else None
        ch = C.byref(C.c_int32(channel)) if channel is not None else None
        check(self._L.rm_enqueue_tx(self._h, src, start_us, air_us, tp, ch))
        self._n_new += 1

    def enqueue_records(self, recs):
        recs = np.ascontiguousarray(recs, dtype=TX_RECORD_DTYPE)
        check(self._L.rm_enqueue_tx_records(self._h, recs.ctypes.data, len(recs)))
        self._n_new += len(recs)

    def tick_flush(self, cap=None):
        n_new = self._n_new
        cap = cap if cap is not None else max(1, n_new) * max(1, self.n)
        cap = min(cap, 1 << 26)
        pkt = np.empty(cap, dtype=np.int32)
        dst = np.empty(cap, dtype=np.int32)
        verdict = np.empty(cap, dtype=np.uint8)
        rssi = np.empty(cap, dtype=np.float64)
        sinr = np.empty(cap, dtype=np.float64)
        pint = np.zeros(max(1, n_new), dtype=np.uint8)
        poff = np.zeros(n_new + 1, dtype=np.uint32)
        cnt = C.c_uint32()
        check(self._L.rm_tick_flush(self._h, pkt.ctypes.data, dst.ctypes.data, verdict.ctypes.data, rssi.ctypes.data,
                                    sinr.ctypes.data, cap, C.byref(cnt), pint.ctypes.data, poff.ctypes.data))
        k = cnt.value
        return TickResult(k, pkt[:k], dst[:k], verdict[:k], rssi[:k], sinr[:k], pint[:n_new], poff)

    @staticmethod
    def _wrap_host_result(r):
        return HostTickView(r)

    def batch_result_view(self, n_slots, raise_on_error=True):
        """Results of slots 0..n_slots-1 of the last batch, wrapped in place in the engine's pinned host
        block (one packing launch, one wait).  Returns (results, status per slot)."""
        res = (HostResult * n_slots)()
        status = (C.c_int32 * n_slots)()
        rc = self._L.rm_batch_result_view(self._h, n_slots, res, status)
        if rc != 0 and (raise_on_error or all(s == 0 for s in status)):
            check(rc)
        return [self._wrap_host_result(r) for r in res], list(status)

    def tick_flush_view(self):
        """Evaluate the enqueued tick; the result stays in the engine's pinned host block and is wrapped,
        not copied (valid until the next evaluating call on this engine)."""
        n_new = self._n_new
        r = HostResult()
        check(self._L.rm_tick_flush_view(self._h, C.byref(r)))
        assert r.n_packets == n_new
        return self._wrap_host_result(r)

    def tick_run(self):
        """Evaluate the enqueued tick; results stay on the device (result_copy / result_device)."""
        check(self._L.rm_tick_run(self._h))

    def draws_pending(self):
        return bool(self._L.rm_draws_pending(self._h))

    def draw_counts_device(self):
        p, n = C.c_void_p(), C.c_int32()
        check(self._L.rm_draw_counts_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def draw_counts_to(self, dev_out_ptr):
        check(self._L.rm_draw_counts_to(self._h, C.c_void_p(dev_out_ptr)))

    def finish_draws(self, all_counts, world, rank):
        """all_counts: host uint32 array [world, n_new] or a device pointer (int)."""
        if isinstance(all_counts, int):
            check(self._L.rm_tick_finish_draws(self._h, C.c_void_p(all_counts), world, rank, 1))
        else:
            a = np.ascontiguousarray(all_counts, dtype=np.uint32)
            check(self._L.rm_tick_finish_draws(self._h, a.ctypes.data, world, rank, 0))

    def draw_nodes_device(self):
        """device pointer of the node index of every link of the last tick that will draw (spatial partitions)"""
        p = C.c_void_p()
        check(self._L.rm_draw_nodes_device(self._h, C.byref(p)))
        return p.value

    def finish_draws_nodes(self, all_counts, all_nodes, stride, world):
        """spatial partitions: all_counts [world, n_new] uint32 and all_nodes [world, stride] int32, host arrays or device
        pointers (both ints)"""
        if isinstance(all_counts, int):
            check(self._L.rm_tick_finish_draws_nodes(self._h, C.c_void_p(all_counts), C.c_void_p(all_nodes), stride, world, 1))
        else:
            a = np.ascontiguousarray(all_counts, dtype=np.uint32)
            b = np.ascontiguousarray(all_nodes, dtype=np.int32)
            check(self._L.rm_tick_finish_draws_nodes(self._h, a.ctypes.data, b.ctypes.data, stride, world, 0))

    def tick(self, recs, t_begin=0, t_end=0, cap=None):
        self.tick_begin(t_begin, t_end)
        self.enqueue_records(recs)
        return self.tick_flush(cap)

    # -- device-resident path
    def pack_tx_device(self, dev_src_ptr, n, start_us, air_us, dev_out_ptr):
        check(self._L.rm_pack_tx_device(self._h, C.c_void_p(dev_src_ptr), n, start_us, air_us,
                                        C.c_void_p(dev_out_ptr)))

    def pack_tx_device_on(self, stream_ptr, dev_src_ptr, n, start_us, air_us, dev_out_ptr):
        check(self._L.rm_pack_tx_device_on(self._h, C.c_void_p(stream_ptr), C.c_void_p(dev_src_ptr), n, start_us, air_us,
                                           C.c_void_p(dev_out_ptr)))

    def pack_tx_batch_device_on(self, stream_ptr, dev_src_ptr, n_ticks, n, start_us, air_us, dev_out_ptr):
        st = np.ascontiguousarray(start_us, dtype=np.int64)
        assert len(st) == n_ticks
        check(self._L.rm_pack_tx_batch_device_on(self._h, C.c_void_p(stream_ptr), C.c_void_p(dev_src_ptr), n_ticks, n,
                                                 st.ctypes.data, air_us, C.c_void_p(dev_out_ptr)))

    def tick_run_device(self, t_begin, t_end, dev_new_ptr, n_new):
        check(self._L.rm_tick_run_device(self._h, t_begin, t_end, C.c_void_p(dev_new_ptr), n_new))

    def tick_run_records_device(self, t_begin, t_end, dev_new_ptr, n_new, latest_end_us):
        """records in device memory for the SINR medium, whose frames stay on the air: `latest_end_us` bounds start + air"""
        check(self._L.rm_tick_run_records_device(self._h, int(t_begin), int(t_end), C.c_void_p(dev_new_ptr), int(n_new), int(latest_end_us)))

    def tick_run_sources_device(self, t_begin, t_end, dev_src_ptr, n, start_us, air_us):
        check(self._L.rm_tick_run_sources_device(self._h, t_begin, t_end, C.c_void_p(dev_src_ptr), n, start_us, air_us))

    # -- several independent ticks per pass (rm_batch_*)
    def batch_run_sources_device(self, t_begin, t_end, dev_src_ptrs, n_src, start_us, air_us):
        """Tick b: sources dev_src_ptrs[b] (device int32[n_src[b]]), frames start at start_us[b]."""
        n = len(dev_src_ptrs)
        i64 = lambda v: np.ascontiguousarray(v, dtype=np.int64)
        tb, te, st, ai = i64(t_begin), i64(t_end), i64(start_us), i64(air_us)
        ptrs = np.ascontiguousarray(dev_src_ptrs, dtype=np.uint64)
        cnt = np.ascontiguousarray(n_src, dtype=np.int32)
        assert len(tb) == len(te) == len(st) == len(ai) == len(cnt) == n
        check(self._L.rm_batch_run_sources_device(self._h, n, tb.ctypes.data, te.ctypes.data, ptrs.ctypes.data,
                                                  cnt.ctypes.data, st.ctypes.data, ai.ctypes.data))

    def batch_run_device(self, t_begin, t_end, dev_rec_ptrs, n_new):
        n = len(dev_rec_ptrs)
        tb = np.ascontiguousarray(t_begin, dtype=np.int64)
        te = np.ascontiguousarray(t_end, dtype=np.int64)
        ptrs = np.ascontiguousarray(dev_rec_ptrs, dtype=np.uint64)
        cnt = np.ascontiguousarray(n_new, dtype=np.int32)
        assert len(tb) == len(te) == len(cnt) == n
        check(self._L.rm_batch_run_device(self._h, n, tb.ctypes.data, te.ctypes.data, ptrs.ctypes.data, cnt.ctypes.data))

    def batch_run_gathered_device(self, t_begin, t_end, dev_gathered_ptr, world, slots):
        """the ticks' records where an all-gather of per-rank blocks left them: [rank][tick][slot]"""
        tb = np.ascontiguousarray(t_begin, dtype=np.int64)
        te = np.ascontiguousarray(t_end, dtype=np.int64)
        check(self._L.rm_batch_run_gathered_device(self._h, len(tb), tb.ctypes.data, te.ctypes.data, C.c_void_p(dev_gathered_ptr),
                                                   world, slots))

    def batch_run_gathered_sources_device(self, t_begin, t_end, dev_src_all_ptr, world, slots, start_us, air_us):
        """the ticks' source indices where an all-gather of per-rank blocks left them: [rank][tick][slot] (-1: padding)"""
        tb = np.ascontiguousarray(t_begin, dtype=np.int64)
        te = np.ascontiguousarray(t_end, dtype=np.int64)
        st = np.ascontiguousarray(start_us, dtype=np.int64)
        check(self._L.rm_batch_run_gathered_sources_device(self._h, len(tb), tb.ctypes.data, te.ctypes.data, C.c_void_p(dev_src_all_ptr),
                                                           world, slots, st.ctypes.data, int(air_us)))

    GATHER_TRAILER = 4   # RM_GATHER_TRAILER: words behind a rank's source indices in its block (digest low, high, two spare)

    def batch_tile_reuse(self):
        """ticks of the last batch a filter workgroup swept with one load of its receivers"""
        return int(self._L.rm_batch_tile_reuse(self._h))

    def table_digest(self):
        """rm_table_digest: a function of the node table's content as this context holds it"""
        d = C.c_uint64(0)
        check(self._L.rm_table_digest(self._h, C.byref(d)))
        return int(d.value)

    @staticmethod
    def gather_blocks(per_rank_sources, digests):
        """what the library's all-gather of a sharded batch delivers: per rank its [ticks][slots] source indices, then the
        trailer with the rank's table digest -> int32 [world][ticks * slots + GATHER_TRAILER]"""
        rows = []
        for src, dg in zip(per_rank_sources, digests):
            flat = np.ascontiguousarray(src, dtype=np.int32).reshape(-1)
            tr = np.array([dg & 0xFFFFFFFF, (dg >> 32) & 0xFFFFFFFF, 0, 0], dtype=np.uint32).view(np.int32)
            rows.append(np.concatenate([flat, tr]))
        return np.stack(rows)

    def batch_run_gathered_blocks_device(self, t_begin, t_end, dev_blocks_ptr, world, slots, start_us, air_us):
        """as batch_run_gathered_sources_device with every rank's block followed by its trailer (gather_blocks): the ranks'
        node-table digests are compared with this context's on the device"""
        tb = np.ascontiguousarray(t_begin, dtype=np.int64)
        te = np.ascontiguousarray(t_end, dtype=np.int64)
        st = np.ascontiguousarray(start_us, dtype=np.int64)
        check(self._L.rm_batch_run_gathered_blocks_device(self._h, len(tb), tb.ctypes.data, te.ctypes.data, C.c_void_p(dev_blocks_ptr),
                                                          world, slots, st.ctypes.data, int(air_us)))

    def prepared(self, name, *args):
        """A call with its arguments converted once: `name` is an entry point that takes the context first; numpy arrays are
        passed by address (and kept alive by the closure).  The returned function costs one ctypes call -- what a host
        loop that issues the same shape of call thousands of times wants (bench.py's timed region)."""
        fn = getattr(self._L, name)
        keep = [np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a for a in args]
        conv = [C.c_void_p(a.ctypes.data) if isinstance(a, np.ndarray) else a for a in keep]
        h = self._h

        def call(_fn=fn, _h=h, _conv=tuple(conv), _keep=keep):
            rc = _fn(_h, *_conv)
            if rc != 0:
                check(rc)
        return call

    # -- RCCL inside the library (rm_comm_*, rm_dist_*)
    @staticmethod
    def comm_available():
        return bool(_lib.lib().rm_comm_available())

    @staticmethod
    def comm_unique_id():
        buf = np.zeros(128, dtype=np.uint8)
        check(_lib.lib().rm_comm_get_unique_id(buf.ctypes.data))
        return buf

    def comm_init_rank(self, unique_id, world, rank):
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.size == 128
        check(self._L.rm_comm_init_rank(self._h, uid.ctypes.data, world, rank))

    def comm_destroy(self):
        check(self._L.rm_comm_destroy(self._h))

    def dist_batch_run_sources_device(self, t_begin, t_end, dev_src_ptr, slots, start_us, air_us):
        """one batch of a receiver-sharded run: pack this rank's transmitters (rows of `slots` node indices), RCCL
        all-gather, sweep -- one call"""
        tb = np.ascontiguousarray(t_begin, dtype=np.int64)
        te = np.ascontiguousarray(t_end, dtype=np.int64)
        st = np.ascontiguousarray(start_us, dtype=np.int64)
        check(self._L.rm_dist_batch_run_sources_device(self._h, len(tb), tb.ctypes.data, te.ctypes.data, C.c_void_p(dev_src_ptr),
                                                       slots, st.ctypes.data, int(air_us)))

    def dist_tick_run_sources_device(self, t_begin, t_end, dev_src_ptr, slots, start_us, air_us):
        check(self._L.rm_dist_tick_run_sources_device(self._h, int(t_begin), int(t_end), C.c_void_p(dev_src_ptr), slots,
                                                      int(start_us), int(air_us)))

    def batch_result_device(self, slot):
        r = DeviceResult()
        check(self._L.rm_batch_result_device(self._h, slot, C.byref(r)))
        return r

    def batch_result_count(self, slot):
        cnt, dropped = C.c_uint32(), C.c_uint32()
        check(self._L.rm_batch_result_count(self._h, slot, C.byref(cnt), C.byref(dropped)))
        return cnt.value, dropped.value

    def batch_result_copy(self, slot, n_new, cap=None):
        """Heard links of tick `slot` of the last batch, copied to the host."""
        cap = cap if cap is not None else min(max(1, n_new) * max(1, self.n), 1 << 26)
        pkt = np.empty(cap, dtype=np.int32)
        dst = np.empty(cap, dtype=np.int32)
        verdict = np.empty(cap, dtype=np.uint8)
        rssi = np.empty(cap, dtype=np.float64)
        sinr = np.empty(cap, dtype=np.float64)
        pint = np.zeros(max(1, n_new), dtype=np.uint8)
        poff = np.zeros(n_new + 1, dtype=np.uint32)
        cnt = C.c_uint32()
        check(self._L.rm_batch_result_copy(self._h, slot, pkt.ctypes.data, dst.ctypes.data, verdict.ctypes.data,
                                           rssi.ctypes.data, sinr.ctypes.data, cap, C.byref(cnt), pint.ctypes.data,
                                           poff.ctypes.data))
        k = cnt.value
        return TickResult(k, pkt[:k], dst[:k], verdict[:k], rssi[:k], sinr[:k], pint[:n_new], poff)

    def result_dense(self):
        """rm_result_dense: the dense tick's heard links as lane masks per (packet, 1024 nodes) cell -- device pointers"""
        r = _lib.DenseResult()
        check(self._L.rm_result_dense(self._h, C.byref(r)))
        return r

    def result_device(self):
        r = DeviceResult()
        check(self._L.rm_result_device(self._h, C.byref(r)))
        return r

    def result_copy(self, n_new, cap=None):
        """Heard links of the last device-path tick, copied to the host."""
        cap = cap if cap is not None else min(max(1, n_new) * max(1, self.n), 1 << 26)
        pkt = np.empty(cap, dtype=np.int32)
        dst = np.empty(cap, dtype=np.int32)
        verdict = np.empty(cap, dtype=np.uint8)
        rssi = np.empty(cap, dtype=np.float64)
        sinr = np.empty(cap, dtype=np.float64)
        pint = np.zeros(max(1, n_new), dtype=np.uint8)
        poff = np.zeros(n_new + 1, dtype=np.uint32)
        cnt = C.c_uint32()
        check(self._L.rm_result_copy(self._h, pkt.ctypes.data, dst.ctypes.data, verdict.ctypes.data, rssi.ctypes.data,
                                     sinr.ctypes.data, cap, C.byref(cnt), pint.ctypes.data, poff.ctypes.data))
        k = cnt.value
        return TickResult(k, pkt[:k], dst[:k], verdict[:k], rssi[:k], sinr[:k], pint[:n_new], poff)

    def result_count(self):
        cnt, dropped = C.c_uint32(), C.c_uint32()
        check(self._L.rm_result_count(self._h, C.byref(cnt), C.byref(dropped)))
        return cnt.value, dropped.value

    # ---- reception stage (Simulator.generate*Events + processAllEvents + Transciever state on the device)
    def events_enable(self, max_packets=0, max_links=0):
        check(self._L.rm_events_enable(self._h, max_packets, max_links))

    def events_disable(self):
        check(self._L.rm_events_disable(self._h))

    def events_next_packet(self):
        return self._L.rm_events_next_packet(self._h)

    def events_process(self, time_us, copy=True, runs=False):
        """Simulator.emulatorTimeStepDone: currentTime = time; processAllEvents(time).  Returns the deliveries of
        the drain in call order: (packet numbers, destination node indices, rssi, pending packets), copied out of the pinned
        block (copy=False: wrapped in place, valid until the next call).  runs=True: as the engine hands them over -- the
        packet numbers once per run of deliveries: (run packet, run first, run count, destinations, rssi, pending packets)."""
        from ._lib import DeliveryView
        v = DeliveryView()
        check(self._L.rm_events_process(self._h, int(time_us), C.byref(v)))
        k = v.count
        self.oldest_pending_packet = v.oldest_packet   # every packet below it has fired its last event

        def arr(ptr, dtype, count=k):
            a = _wrap(ptr, dtype, count)
            return a.copy() if copy else a
        # the packet numbers come once per run of deliveries (a packet's deliveries are adjacent): spread out here
        n_runs = v.n_runs
        if runs:
            return (arr(v.run_packet, np.int64, n_runs), arr(v.run_first, np.uint32, n_runs), arr(v.run_count, np.uint32, n_runs),
                    arr(v.dst, np.int32), arr(v.rssi, np.float64), v.pending_packets)
        first, cnt = _wrap(v.run_first, np.uint32, n_runs), _wrap(v.run_count, np.uint32, n_runs)
        assert n_runs == 0 or (first[0] == 0 and int(first[-1]) + int(cnt[-1]) == k and np.all(first[1:] == (first[:-1] + cnt[:-1])))
        packet = np.repeat(_wrap(v.run_packet, np.int64, n_runs), cnt)
        return packet, arr(v.dst, np.int32), arr(v.rssi, np.float64), v.pending_packets

    def node_info(self, nodes=None, n=None):
        """(rssi, receiving state, channel) per node: the node-info of a time-step message."""
        if nodes is not None:
            nodes = np.ascontiguousarray(nodes, dtype=np.int32)
            n = len(nodes)
        elif n is None:
            n = self._L.rm_node_count(self._h)
        rssi = np.empty(n, dtype=np.float64)
        rx = np.empty(n, dtype=np.int32)
        ch = np.empty(n, dtype=np.int32)
        check(self._L.rm_node_info(self._h, nodes.ctypes.data if nodes is not None else None, n, rssi.ctypes.data,
                                   rx.ctypes.data, ch.ctypes.data))
        return rssi, rx, ch

    def node_info_changed(self):
        """(nodes, rssi, receiving, channel) of the nodes whose node-info differs from what this call reported for them last"""
        n = max(self.n, 1)
        nodes = np.empty(n, dtype=np.int32)
        rssi = np.empty(n, dtype=np.float64)
        rx = np.empty(n, dtype=np.int32)
        ch = np.empty(n, dtype=np.int32)
        k = C.c_int32(0)
        check(self._L.rm_node_info_changed(self._h, nodes.ctypes.data, rssi.ctypes.data, rx.ctypes.data, ch.ctypes.data, n, C.byref(k)))
        return nodes[:k.value], rssi[:k.value], rx[:k.value], ch[:k.value]

    def sync(self):
        check(self._L.rm_sync(self._h))

    def profile_enable(self, every_n=1):
        check(self._L.rm_profile_enable(self._h, int(every_n)))

    STAGES = ("k_filter", "k_exact", "k_self_entries", "k_cell_off+k_slot_scan", "k_sinr", "k_finalize",
              "k_reorder", "draw kernels", "empty bracket")

    def profile_read(self):
        """-> (sampled ticks, {stage name: summed milliseconds})"""
        n = C.c_uint32()
        ms = (C.c_double * 9)()
        check(self._L.rm_profile_read(self._h, C.byref(n), ms))
        return n.value, {name: ms[i] for i, name in enumerate(self.STAGES)}

    def profile_kernels(self):
        """-> {kernel name as rocprofv3 prints it: (sampled launches, summed milliseconds of the kernel's own dispatch intervals, stage)}"""
        from ._lib import KernelTime
        n = C.c_int32(0)
        buf = (KernelTime * 64)()
        check(self._L.rm_profile_kernels(self._h, buf, 64, C.byref(n)))
        return {buf[i].name.decode(): (buf[i].launches, buf[i].total_ms, self.STAGES[buf[i].stage]) for i in range(min(n.value, 64))}

    def slot_stats(self, slot=0):
        """(candidate links of the sweep's filter, heard links) of result slot `slot`; synchronises"""
        cand, heard = C.c_uint64(0), C.c_uint64(0)
        check(self._L.rm_slot_stats(self._h, slot, C.byref(cand), C.byref(heard)))
        return cand.value, heard.value

    def air_list_stats(self):
        """(ticks that only added their new frames to the on-air lists, ticks that rebuilt the lists) -- SINR extension"""
        inc, reb = C.c_uint64(0), C.c_uint64(0)
        check(self._L.rm_air_list_stats(self._h, C.byref(inc), C.byref(reb)))
        return inc.value, reb.value

    def air_scan_ticks(self):
        """ticks of the SINR extension evaluated by scan: interferers found among the frames on the air, no lists"""
        v = C.c_uint64(0)
        check(self._L.rm_air_scan_ticks(self._h, C.byref(v)))
        return v.value

    def air_batch_stats(self):
        """(batches, ticks) of the SINR extension whose frames outlived their tick and that were swept as batches (rm_airbatch.hip)"""
        b, t = C.c_uint64(0), C.c_uint64(0)
        check(self._L.rm_air_batch_stats(self._h, C.byref(b), C.byref(t)))
        return b.value, t.value

    def air_batch_pairs(self):
        """(pairs evaluated exactly, frames indexed, pairs that interfered) of the last batch of overlapping SINR ticks; synchronises"""
        p, f, i = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(self._L.rm_air_batch_pairs(self._h, C.byref(p), C.byref(f), C.byref(i)))
        return p.value, f.value, i.value

    def air_ring_stats(self):
        """(entries allocated in the busiest sub-ring since the lists were last rebuilt, entries a sub-ring holds)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self._L.rm_air_ring_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_link_evaluations(self):
        return self._L.rm_last_link_evaluations(self._h)


class Group:
    """n engine contexts behind one caller (rm_group_*): receivers range-partitioned over the members, a tick's Tx
    records handed to every member from the host, results merged in node order."""

    def __init__(self, devices, spatial=True):
        self._L = _lib.lib()
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        check(self._L.rm_group_create(len(devices), devs, C.byref(h)))
        self._h = h
        self._n_new = 0
        check(self._L.rm_group_set_partitioning(self._h, 1 if spatial else 0))   # regions of the plane / index ranges

    def close(self):
        if self._h:
            self._L.rm_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return self._L.rm_group_size(self._h)

    def set_model(self, kind, **kw):
        p = Engine.default_params(kind)
        for k, v in kw.items():
            assert hasattr(p, k), k
            setattr(p, k, v)
        check(self._L.rm_group_set_model(self._h, C.byref(p)))

    def seed(self, seed):
        check(self._L.rm_group_seed(self._h, seed))

    @property
    def rng_state(self):
        st = C.c_uint64(0)
        check(self._L.rm_group_get_rng_state(self._h, C.byref(st)))
        return st.value

    def set_link_capacity(self, cap):
        check(self._L.rm_group_set_link_capacity(self._h, cap))

    def upload_table(self, nd):
        def arr(a, dt):
            return np.ascontiguousarray(a, dtype=dt)
        self._keep = [arr(nd.x, np.float64), arr(nd.y, np.float64), arr(nd.z, np.float64), arr(nd.txpower, np.float64),
                      arr(nd.channel, np.int32), arr(nd.enabled, np.uint8), arr(nd.rxprob, np.float64),
                      arr(nd.txprob, np.float64), arr(nd.int_id, np.int32)]
        check(self._L.rm_group_nodes_upload(self._h, nd.n, *[a.ctypes.data for a in self._keep]))
        self._n = nd.n

    def uses_rccl(self):
        rc = self._L.rm_group_uses_rccl(self._h)
        if rc < 0:
            check(rc)
        return bool(rc)

    def tick_run_sources_device(self, t_begin, t_end, dev_src_ptrs, slots, start_us, air_us):
        """the device-resident tick: dev_src_ptrs[r] = `slots` source indices in member r's device memory"""
        ptrs = (C.c_void_p * len(dev_src_ptrs))(*[int(p) for p in dev_src_ptrs])
        check(self._L.rm_group_tick_run_sources_device(self._h, int(t_begin), int(t_end), ptrs, slots, int(start_us), int(air_us)))
        self._n_new = slots * len(dev_src_ptrs)

    def result_copy(self, cap=None):
        n_new = self._n_new
        if cap is None:
            cap = max(1, n_new) * max(1, self._n)
        pkt = np.empty(cap, dtype=np.int32)
        dst = np.empty(cap, dtype=np.int32)
        verdict = np.empty(cap, dtype=np.uint8)
        rssi = np.empty(cap, dtype=np.float64)
        sinr = np.empty(cap, dtype=np.float64)
        pint = np.zeros(max(1, n_new), dtype=np.uint8)
        poff = np.zeros(n_new + 1, dtype=np.uint32)
        cnt = C.c_uint32(0)
        check(self._L.rm_group_result_copy(self._h, pkt.ctypes.data, dst.ctypes.data, verdict.ctypes.data, rssi.ctypes.data,
                                           sinr.ctypes.data, cap, C.byref(cnt), pint.ctypes.data, poff.ctypes.data))
        k = cnt.value
        return TickResult(k, pkt[:k], dst[:k], verdict[:k], rssi[:k], sinr[:k], pint[:n_new], poff)

    def tick(self, recs, t_begin=0, t_end=0, cap=None):
        recs = np.ascontiguousarray(recs, dtype=TX_RECORD_DTYPE)
        n_new = len(recs)
        check(self._L.rm_group_tick_begin(self._h, t_begin, t_end))
        check(self._L.rm_group_enqueue_tx_records(self._h, recs.ctypes.data, n_new))
        if cap is None:
            cap = max(1, n_new) * max(1, self._n)
        pkt = np.empty(cap, dtype=np.int32)
        dst = np.empty(cap, dtype=np.int32)
        verdict = np.empty(cap, dtype=np.uint8)
        rssi = np.empty(cap, dtype=np.float64)
        sinr = np.empty(cap, dtype=np.float64)
        pint = np.zeros(max(1, n_new), dtype=np.uint8)
        poff = np.zeros(n_new + 1, dtype=np.uint32)
        cnt = C.c_uint32(0)
        check(self._L.rm_group_tick_flush(self._h, pkt.ctypes.data, dst.ctypes.data, verdict.ctypes.data, rssi.ctypes.data,
                                          sinr.ctypes.data, cap, C.byref(cnt), pint.ctypes.data, poff.ctypes.data))
        k = cnt.value
        return TickResult(k, pkt[:k], dst[:k], verdict[:k], rssi[:k], sinr[:k], pint[:n_new], poff)
