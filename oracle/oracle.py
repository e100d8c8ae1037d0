"""ctypes binding of the CPU oracle (oracle/rm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librm_oracle.so")

MODEL_NULL, MODEL_UDGM, MODEL_UDGM_CONST, MODEL_N2N, MODEL_LOGDIST = range(5)
UNHEARD, INTERFERED, DELIVERED = 0, 1, 2
LD_SINR = 1


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("rm_oracle.c", "rm_events.c", "rm_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "librm_oracle.so"])
    return _SO


class Nodes(C.Structure):
    _fields_ = [("n", C.c_int32),
                ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("txpower", C.c_void_p),
                ("channel", C.c_void_p), ("enabled", C.c_void_p),
                ("rxprob", C.c_void_p), ("txprob", C.c_void_p), ("int_id", C.c_void_p)]


class Model(C.Structure):
    _fields_ = [("kind", C.c_int32),
                ("udgm_success_ratio_tx", C.c_double), ("udgm_success_ratio_rx", C.c_double),
                ("udgm_transmission_range", C.c_double), ("udgm_interference_range", C.c_double),
                ("const_range", C.c_double),
                ("n2n_matrix", C.c_void_p), ("n2n_m", C.c_int32),
                ("ld_pl0_db", C.c_double), ("ld_exponent", C.c_double), ("ld_d0", C.c_double),
                ("ld_sigma_db", C.c_double), ("ld_clip", C.c_double), ("ld_seed", C.c_uint64),
                ("ld_sensitivity_dbm", C.c_double), ("ld_noise_dbm", C.c_double),
                ("ld_capture_db", C.c_double), ("ld_ifloor_dbm", C.c_double),
                ("ld_flags", C.c_int32)]


class Packet(C.Structure):
    _fields_ = [("src", C.c_int32), ("channel", C.c_int32),
                ("x", C.c_double), ("y", C.c_double), ("z", C.c_double),
                ("txpower", C.c_double), ("txprob", C.c_double),
                ("start_us", C.c_int64), ("air_us", C.c_int64)]


PACKET_DTYPE = np.dtype([("src", "<i4"), ("channel", "<i4"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8"),
                         ("txpower", "<f8"), ("txprob", "<f8"), ("start_us", "<i8"), ("air_us", "<i8")])
assert PACKET_DTYPE.itemsize == C.sizeof(Packet)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_model_defaults.argtypes = [C.POINTER(Model), C.c_int32]
        L.orc_jrandom_seed.restype = C.c_uint64
        L.orc_jrandom_seed.argtypes = [C.c_int64]
        L.orc_jrandom_next_int.restype = C.c_int32
        L.orc_jrandom_next_int.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_jrandom_next_double.restype = C.c_double
        L.orc_jrandom_next_double.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_distance.restype = C.c_double
        L.orc_distance.argtypes = [C.c_double] * 6
        L.orc_air_time_us.restype = C.c_int64
        L.orc_air_time_us.argtypes = [C.c_int64]
        L.orc_event_times.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        for name in ("orc_udgm_rx_probability", "orc_n2n_rx_probability", "orc_logdist_rssi"):
            f = getattr(L, name)
            f.restype = C.c_double
        L.orc_udgm_rx_probability.argtypes = [C.POINTER(Model), C.POINTER(Nodes), C.POINTER(Packet), C.c_int32]
        L.orc_n2n_rx_probability.argtypes = [C.POINTER(Model), C.POINTER(Nodes), C.POINTER(Packet), C.c_int32]
        L.orc_logdist_rssi.argtypes = [C.POINTER(Model), C.POINTER(Packet), C.POINTER(Nodes), C.c_int32]
        L.orc_udgm_tx_probability.restype = C.c_double
        L.orc_udgm_tx_probability.argtypes = [C.POINTER(Model), C.POINTER(Packet)]
        L.orc_tick.restype = C.c_int64
        L.orc_tick.argtypes = [C.POINTER(Model), C.POINTER(Nodes), C.POINTER(C.c_uint64),
                               C.c_void_p, C.c_int32, C.c_int32,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_void_p]
        L.orc_tick_mt.restype = C.c_int64
        L.orc_tick_mt.argtypes = [C.POINTER(Model), C.POINTER(Nodes), C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_udgm_pow_sensitivity.restype = None
        L.orc_udgm_pow_sensitivity.argtypes = [C.POINTER(Model), C.POINTER(Nodes), C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                               C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.orc_count_links.restype = C.c_int64
        L.orc_count_links.argtypes = [C.POINTER(Model), C.POINTER(Nodes), C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_int32, C.POINTER(C.c_int64)]
        L.orc_max_threads.restype = C.c_int32
        for name in ("orc_det_log2", "orc_det_exp2", "orc_det_log10", "orc_det_pow10", "orc_det_normal",
                     "orc_fixed_roundtrip"):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.c_double]
        L.orc_shadow_hash.restype = C.c_uint64
        L.orc_shadow_hash.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_shadow_gauss.restype = C.c_double
        L.orc_shadow_gauss.argtypes = [C.POINTER(Model), C.c_uint32, C.c_uint32]
        # serial replay of the event queue / tick-end drain / receiver state machine (rm_events.c)
        L.orc_sim_create.restype = C.c_void_p
        L.orc_sim_create.argtypes = [C.c_int32]
        L.orc_sim_destroy.argtypes = [C.c_void_p]
        for name in ("orc_sim_time", "orc_sim_move_tops", "orc_sim_top_start", "orc_sim_pending"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_sim_error.restype = C.c_int32
        L.orc_sim_error.argtypes = [C.c_void_p]
        L.orc_sim_transmission_events.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64]
        L.orc_sim_reception_events.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_double, C.c_int32]
        L.orc_sim_step.restype = C.c_int64
        L.orc_sim_step.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.orc_sim_rssi.restype = C.c_double
        L.orc_sim_rssi.argtypes = [C.c_void_p, C.c_int32, C.c_double]
        L.orc_sim_receiving_state.restype = C.c_int32
        L.orc_sim_receiving_state.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.orc_sim_receiving_packet.restype = C.c_int32
        L.orc_sim_receiving_packet.argtypes = [C.c_void_p, C.c_int32]
        L.orc_sim_sending_packet.restype = C.c_int32
        L.orc_sim_sending_packet.argtypes = [C.c_void_p, C.c_int32]
        L.orc_sim_medium_calls.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int32]
        L.orc_evq_replay.restype = C.c_int64
        L.orc_evq_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        _lib = L
    return _lib


EV_RX_START, EV_RX_END_INTERFERENCE, EV_RX_END_DELIVERY, EV_TX_START, EV_TX_END = range(5)
EVENT_DTYPE = np.dtype([("time", "<i8"), ("node", "<i4"), ("pkt", "<i4"), ("kind", "<i4"), ("pad", "<i4"), ("rssi", "<f8")])
assert EVENT_DTYPE.itemsize == 32


class Sim:
    """Serial replay of what consumes the verdicts: the reference's ladder queue (EventQueue.java, literal),
    Simulator.generate*Events / processAllEvents and the Transciever state machine (oracle/rm_events.c)."""

    def __init__(self, n_nodes):
        self._L = lib()
        self.n = n_nodes
        self._h = C.c_void_p(self._L.orc_sim_create(n_nodes))

    def close(self):
        if self._h:
            self._L.orc_sim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def time(self):
        return self._L.orc_sim_time(self._h)

    @property
    def error(self):
        return self._L.orc_sim_error(self._h)

    @property
    def move_tops(self):
        return self._L.orc_sim_move_tops(self._h)

    @property
    def top_start(self):
        return self._L.orc_sim_top_start(self._h)

    @property
    def pending(self):
        return self._L.orc_sim_pending(self._h)

    def transmission_events(self, pkt, src, start_us, air_us):
        self._L.orc_sim_transmission_events(self._h, pkt, src, start_us, air_us)

    def reception_events(self, pkt, dst, start_us, air_us, rssi, deliver):
        self._L.orc_sim_reception_events(self._h, pkt, dst, start_us, air_us, float(rssi), 1 if deliver else 0)

    def medium_calls(self, res, packets, pkt_base=0, const_loss=False):
        """What a reference medium does with one evaluated tick `res` (oracle TickResult) of `packets`, in the
        reference's order: per packet generateTransmissionEvents, then generateReceptionEvents per heard receiver in
        node order (UDGMRadioMedium.java:97-111).  Returns the immediate deliveries of the constant-loss medium
        (UDGMConstantLossRadioMedium.java:30: deliverRadioPacket, no events)."""
        packets = np.ascontiguousarray(np.atleast_1d(packets), dtype=PACKET_DTYPE)
        pkt = np.ascontiguousarray(res.pkt, dtype=np.int32)
        dst = np.ascontiguousarray(res.dst, dtype=np.int32)
        verdict = np.ascontiguousarray(res.verdict, dtype=np.uint8)
        rssi = np.ascontiguousarray(res.rssi, dtype=np.float64)
        self._L.orc_sim_medium_calls(self._h, packets.ctypes.data, len(packets), pkt_base, len(pkt), pkt.ctypes.data,
                                     dst.ctypes.data, verdict.ctypes.data, rssi.ctypes.data, 1 if const_loss else 0)
        if const_loss:
            return [(pkt_base + int(q), int(d), float(r)) for q, d, r in zip(pkt, dst, rssi)]
        return []

    def step(self, time_us, cap=None):
        """emulatorTimeStepDone: currentTime = time; processAllEvents(time).  Executed events in pop order."""
        if cap is None:
            cap = max(1, self.pending)
        ev = np.zeros(cap, dtype=EVENT_DTYPE)
        n = self._L.orc_sim_step(self._h, time_us, ev.ctypes.data, cap)
        assert n <= cap
        return ev[:n]

    def rssi(self, node, base_rssi=-100.0):
        return self._L.orc_sim_rssi(self._h, node, base_rssi)

    def receiving_state(self, node, enabled=True):
        return self._L.orc_sim_receiving_state(self._h, node, 1 if enabled else 0)

    def node_info(self, nodes=None, base_rssi=-100.0, enabled=None):
        idx = range(self.n) if nodes is None else nodes
        rssi = np.array([self.rssi(i, base_rssi) for i in idx])
        st = np.array([self.receiving_state(i, True if enabled is None else bool(enabled[i])) for i in idx], dtype=np.int32)
        return rssi, st


def evq_replay(ops):
    """ops: list of ('add', time) / ('pop', time).  Returns the popped events' insertion numbers in pop order."""
    t = np.array([o[1] for o in ops], dtype=np.int64)
    k = np.array([0 if o[0] == "add" else 1 for o in ops], dtype=np.int32)
    n_add = int((k == 0).sum())
    out = np.zeros(max(1, n_add), dtype=np.int64)
    n = lib().orc_evq_replay(t.ctypes.data, k.ctypes.data, len(ops), out.ctypes.data, len(out))
    if n < 0:
        raise RuntimeError("the Java queue would have thrown (code %d)" % -n)
    return out[:n]


class JavaRandom:
    """java.util.Random (seeded), Java SE specification."""

    def __init__(self, seed):
        self.state = C.c_uint64(lib().orc_jrandom_seed(seed))

    def next_int(self):
        return lib().orc_jrandom_next_int(C.byref(self.state))

    def next_double(self):
        return lib().orc_jrandom_next_double(C.byref(self.state))


class NodeTable:
    """Struct-of-arrays node state with the reference's defaults (Transciever.java:11-18)."""

    def __init__(self, n):
        self.n = n
        self.x = np.zeros(n)
        self.y = np.zeros(n)
        self.z = np.zeros(n)
        self.txpower = np.zeros(n)
        self.channel = np.full(n, 26, dtype=np.int32)
        self.enabled = np.ones(n, dtype=np.uint8)
        self.rxprob = np.ones(n)
        self.txprob = np.ones(n)
        self.int_id = np.arange(1, n + 1, dtype=np.int32)

    def as_struct(self):
        for name, dt in (("x", np.float64), ("y", np.float64), ("z", np.float64), ("txpower", np.float64),
                         ("channel", np.int32), ("enabled", np.uint8), ("rxprob", np.float64),
                         ("txprob", np.float64), ("int_id", np.int32)):
            a = np.ascontiguousarray(getattr(self, name), dtype=dt)
            assert a.shape == (self.n,), name
            setattr(self, name, a)
        s = Nodes()
        s.n = self.n
        for name in ("x", "y", "z", "txpower", "channel", "enabled", "rxprob", "txprob", "int_id"):
            setattr(s, name, getattr(self, name).ctypes.data)
        return s

    def packet(self, src, start_us=0, air_us=0, txpower=None, channel=None):
        """RadioPacket(node, time, data): copies txpower/channel from the source (RadioPacket.java:46-52)."""
        p = np.zeros((), dtype=PACKET_DTYPE)
        p["src"] = src
        p["channel"] = self.channel[src] if channel is None else channel
        p["x"], p["y"], p["z"] = self.x[src], self.y[src], self.z[src]
        p["txpower"] = self.txpower[src] if txpower is None else txpower
        p["txprob"] = self.txprob[src]
        p["start_us"] = start_us
        p["air_us"] = air_us
        return p

    def packets(self, srcs, start_us=0, air_us=0):
        srcs = np.asarray(srcs, dtype=np.int32)
        p = np.zeros(len(srcs), dtype=PACKET_DTYPE)
        p["src"] = srcs
        p["channel"] = self.channel[srcs]
        p["x"], p["y"], p["z"] = self.x[srcs], self.y[srcs], self.z[srcs]
        p["txpower"] = self.txpower[srcs]
        p["txprob"] = self.txprob[srcs]
        p["start_us"] = start_us
        p["air_us"] = air_us
        return p


def model(kind, **kw):
    m = Model()
    lib().orc_model_defaults(C.byref(m), kind)
    keep = None
    for k, v in kw.items():
        if k == "n2n_matrix":
            keep = np.ascontiguousarray(v, dtype=np.float64)
            assert keep.ndim == 2 and keep.shape[0] == keep.shape[1]
            m.n2n_matrix = keep.ctypes.data
            m.n2n_m = keep.shape[0]
        else:
            assert hasattr(m, k), k
            setattr(m, k, v)
    m._keep = keep
    return m


class TickResult:
    def __init__(self, count, pkt, dst, verdict, rssi, sinr, pkt_interference, pkt_draws, rng_state):
        self.count = count
        self.pkt, self.dst, self.verdict, self.rssi, self.sinr = pkt, dst, verdict, rssi, sinr
        self.pkt_interference, self.pkt_draws, self.rng_state = pkt_interference, pkt_draws, rng_state


def tick(mdl, nodes, active, first_new=0, rng_state=0, cap=None):
    """Run one oracle pass; `active` is a PACKET_DTYPE array. Returns TickResult (arrays trimmed to count)."""
    L = lib()
    active = np.ascontiguousarray(np.atleast_1d(active), dtype=PACKET_DTYPE)
    n_active = len(active)
    n_new = n_active - first_new
    ns = nodes.as_struct()
    if cap is None:
        cap = max(1, n_new) * max(1, nodes.n)
    pkt = np.empty(cap, dtype=np.int32)
    dst = np.empty(cap, dtype=np.int32)
    verdict = np.empty(cap, dtype=np.uint8)
    rssi = np.empty(cap, dtype=np.float64)
    sinr = np.empty(cap, dtype=np.float64)
    pint = np.zeros(max(1, n_new), dtype=np.uint8)
    pdraw = np.zeros(max(1, n_new), dtype=np.int32)
    st = C.c_uint64(int(rng_state))
    cnt = L.orc_tick(C.byref(mdl), C.byref(ns), C.byref(st), active.ctypes.data, n_active, first_new,
                     pkt.ctypes.data, dst.ctypes.data, verdict.ctypes.data, rssi.ctypes.data, sinr.ctypes.data,
                     cap, pint.ctypes.data, pdraw.ctypes.data)
    k = min(cnt, cap)
    return TickResult(cnt, pkt[:k], dst[:k], verdict[:k], rssi[:k], sinr[:k], pint[:n_new], pdraw[:n_new],
                      st.value)


def tick_mt(mdl, nodes, active, first_new=0, threads=0, cap=None):
    """The pass over `threads` threads (0: all the box offers) -- only for ticks in which no java.util.Random draw happens
    (raises otherwise): the packets are then independent.  Same result object as tick(); rng_state / pkt_draws are 0."""
    L = lib()
    active = np.ascontiguousarray(np.atleast_1d(active), dtype=PACKET_DTYPE)
    n_active = len(active)
    n_new = n_active - first_new
    ns = nodes.as_struct()
    if cap is None:
        cap = max(1, n_new) * max(1, nodes.n)
    pkt = np.empty(cap, dtype=np.int32)
    dst = np.empty(cap, dtype=np.int32)
    verdict = np.empty(cap, dtype=np.uint8)
    rssi = np.empty(cap, dtype=np.float64)
    sinr = np.empty(cap, dtype=np.float64)
    pint = np.zeros(max(1, n_new), dtype=np.uint8)
    cnt = L.orc_tick_mt(C.byref(mdl), C.byref(ns), active.ctypes.data, n_active, first_new, threads or L.orc_max_threads(),
                        pkt.ctypes.data, dst.ctypes.data, verdict.ctypes.data, rssi.ctypes.data, sinr.ctypes.data, cap,
                        pint.ctypes.data)
    if cnt == -2:
        raise ValueError("this tick consumes java.util.Random draws: only the serial pass (tick) is the reference's")
    if cnt < 0:
        raise MemoryError("orc_tick_mt")
    k = min(cnt, cap)
    return TickResult(cnt, pkt[:k], dst[:k], verdict[:k], rssi[:k], sinr[:k], pint[:n_new], np.zeros(max(1, n_new), dtype=np.int32)[:n_new], 0)


def count_links(mdl, nodes, active, first_new=0, threads=1):
    L = lib()
    active = np.ascontiguousarray(np.atleast_1d(active), dtype=PACKET_DTYPE)
    ns = nodes.as_struct()
    deliv = C.c_int64(0)
    heard = L.orc_count_links(C.byref(mdl), C.byref(ns), active.ctypes.data, len(active), first_new, threads,
                              C.byref(deliv))
    return heard, deliv.value


def udgm_pow_sensitivity(mdl, nodes, packets, d2_ulp, dmax2_ulp):
    """TEST-ONLY: UDGMRadioMedium.java:69,74 with Math.pow's results moved by whole ulps against x * x
    -> (links evaluated, heard, heard/unheard flips, heard links whose p differs, largest relative change of p)"""
    L = lib()
    packets = np.ascontiguousarray(np.atleast_1d(packets), dtype=PACKET_DTYPE)
    ns = nodes.as_struct()
    out = (C.c_int64 * 4)()
    rel = C.c_double(0.0)
    L.orc_udgm_pow_sensitivity(C.byref(mdl), C.byref(ns), packets.ctypes.data, len(packets), int(d2_ulp), int(dmax2_ulp), out, C.byref(rel))
    return int(out[0]), int(out[1]), int(out[2]), int(out[3]), float(rel.value)
