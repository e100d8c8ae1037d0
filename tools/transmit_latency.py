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
