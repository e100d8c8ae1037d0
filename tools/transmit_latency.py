"""Latency of the per-packet drop-in call (rm_transmit: host arguments in, heard links out on the
host), the call the Java shim makes once per RadioMedium.transmit, and of one-tick host-buffer
flushes of T frames.  Run on the GPU box:  python tools/transmit_latency.py [nodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import radio_sim_amd as rsa
from radio_sim_amd import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
nodes = W.make_nodes(n, 3)
for name, kind, kw in (("udgm", rsa.MODEL_UDGM, {}), ("logdist_shadow", rsa.MODEL_LOGDIST, W.model_kwargs("logdist_shadow")[1])):
    eng = rsa.Engine(0)
    eng.upload_table(nodes)
    eng.set_model(kind, **kw)
    srcs = np.random.default_rng(1).integers(0, n, 300)
    for s in srcs[:50]:
        eng.transmit(int(s), 0, 254, cap=4096)
    t0 = time.perf_counter()
    heard = 0
    for s in srcs[50:]:
        heard += eng.transmit(int(s), 0, 254, cap=4096).count
    dt = (time.perf_counter() - t0) / 250
    print("%-15s N=%d  rm_transmit: %.1f us per packet (%.1f heard links each)" % (name, n, dt * 1e6, heard / 250))
    eng.close()

# one tick through host buffers: records in (rm_enqueue_tx_records), heard links out (rm_tick_flush)
eng = rsa.Engine(0)
eng.upload_table(nodes)
eng.set_model(rsa.MODEL_LOGDIST, **W.model_kwargs("logdist_shadow")[1])
for T in (10, 100, 1000):
    srcs = np.sort(np.random.default_rng(2).choice(n, T, replace=False))
    recs = np.zeros(T, dtype=rsa.TX_RECORD_DTYPE)
    recs["x"], recs["y"], recs["z"] = nodes.x[srcs], nodes.y[srcs], nodes.z[srcs]
    recs["txpower"], recs["txprob"], recs["channel"] = nodes.txpower[srcs], nodes.txprob[srcs], nodes.channel[srcs]
    recs["src"], recs["air_us"] = srcs, W.AIR_US
    for _ in range(5):
        res = eng.tick(recs, 0, 1000, cap=1 << 17)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        res = eng.tick(recs, 0, 1000, cap=1 << 17)
    dt = (time.perf_counter() - t0) / reps
    print("rm_tick_flush       N=%d  T=%-5d %.1f us per tick, host buffers in and out (%d heard links, %.2f MB out) = %.2e links/s"
          % (n, T, dt * 1e6, res.count, res.count * 25 / 1e6, T * (n - 1) / dt))
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.tick_begin(0, 1000)
        eng.enqueue_records(recs)
        res = eng.tick_flush_view()
    dt = (time.perf_counter() - t0) / reps
    print("rm_tick_flush_view  N=%d  T=%-5d %.1f us per tick, records in, result read in place = %.2e links/s"
          % (n, T, dt * 1e6, T * (n - 1) / dt))
eng.close()

# a batch of ticks with the results brought to the host: rm_batch_run_sources_device + rm_batch_result_view
import ctypes as C
hip = C.CDLL("libamdhip64.so.7")
eng = rsa.Engine(0)
eng.upload_table(nodes)
eng.set_model(rsa.MODEL_LOGDIST, **W.model_kwargs("logdist_shadow")[1])
eng.set_link_capacity(1 << 21)
T, NB = n // 100, 64
ptrs = []
for b in range(NB):
    srcs = np.sort(np.random.default_rng(100 + b).choice(n, T, replace=False)).astype(np.int32)
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), C.c_size_t(srcs.nbytes)) == 0
    assert hip.hipMemcpy(d, C.c_void_p(srcs.ctypes.data), C.c_size_t(srcs.nbytes), 1) == 0
    ptrs.append(d.value)
tb = [1000 * b for b in range(NB)]
te = [t + 1000 for t in tb]
for _ in range(3):
    eng.batch_run_sources_device(tb, te, ptrs, [T] * NB, tb, [W.AIR_US] * NB)
    views, _ = eng.batch_result_view(NB)
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    eng.batch_run_sources_device(tb, te, ptrs, [T] * NB, tb, [W.AIR_US] * NB)
    views, _ = eng.batch_result_view(NB)
dt = (time.perf_counter() - t0) / reps
links = sum(v.count for v in views)
print("batch of %d ticks + rm_batch_result_view  N=%d T=%d: %.1f us per tick with all results on the host (%.1f MB per batch) = %.2e links/s"
      % (NB, n, T, dt / NB * 1e6, links * 25 / 1e6, NB * T * (n - 1) / dt))
eng.close()
