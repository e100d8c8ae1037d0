"""Host time to issue one rm_batch_run_sources_device call of 64 ticks (no synchronisation) against the
device time the batch takes: is the bench's loop bound by the host?  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
import radio_sim_amd as rsa
from radio_sim_amd import workload as W

n, NB = 100_000, 64
T = n // 100
nodes = W.make_nodes(n, 3)
hip = C.CDLL("libamdhip64.so.7")
engs = []
for _ in range(2):
    e = rsa.Engine(0)
    e.upload_table(nodes)
    e.set_model(rsa.MODEL_LOGDIST, **W.model_kwargs("logdist_shadow")[1])
    e.set_link_capacity(1 << 21)
    engs.append(e)
ptrs = []
for b in range(NB):
    srcs = np.sort(np.random.default_rng(100 + b).choice(n, T, replace=False)).astype(np.int32)
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), C.c_size_t(srcs.nbytes)) == 0
    assert hip.hipMemcpy(d, C.c_void_p(srcs.ctypes.data), C.c_size_t(srcs.nbytes), 1) == 0
    ptrs.append(d.value)
tb = np.arange(NB, dtype=np.int64) * 1000
args = (tb, tb + 1000, np.array(ptrs, dtype=np.uint64), np.full(NB, T, dtype=np.int32), tb, np.full(NB, W.AIR_US, dtype=np.int64))
for e in engs:
    e.batch_run_sources_device(*args)
    e.sync()
th = 0.0
for k in range(20):          # the device idle at every call: the host's own cost (descriptors, launches)
    engs[0].sync()
    t0 = time.perf_counter()
    engs[0].batch_run_sources_device(*args)
    th += time.perf_counter() - t0
engs[0].sync()
print("host cost of one call with the device idle: %.1f us (%.2f us per tick)" % (th / 20 * 1e6, th / 20 / NB * 1e6))
for contexts in (1, 2):
    reps = 40
    t0 = time.perf_counter()
    for k in range(reps):
        engs[k % contexts].batch_run_sources_device(*args)
    t_issue = time.perf_counter() - t0
    for e in engs:
        e.sync()
    t_all = time.perf_counter() - t0
    print("%d context(s): %.1f us of host time per call (64 ticks), %.1f us per batch end to end = %.2f us per tick"
          % (contexts, t_issue / reps * 1e6, t_all / reps * 1e6, t_all / reps / NB * 1e6))
