"""Per-tick cost of the reception stage: rm_tick_run_sources_device + rm_events_process (deliveries to the host)
on the BASELINE shapes.  Run on the GPU box:  python tools/events_latency.py [workload] [ticks]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import radio_sim_amd as rsa  # noqa: E402
from radio_sim_amd import workload as W  # noqa: E402
from util import DeviceArray  # noqa: E402

SHAPES = {"c2": (2, 10_000, 0.01, "logdist"), "c3": (3, 100_000, 0.01, "logdist_shadow"), "udgm": (3, 100_000, 0.01, "udgm")}
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 200
idx, n, frac, model = SHAPES[wl]
t = int(round(n * frac))
nodes = W.make_nodes(n, idx)
kind_name, kw = W.model_kwargs(model)
_stream = None
if len(sys.argv) > 3 and sys.argv[3] == "torch":   # as bench.py drives it: the context on a torch stream
    import torch
    torch.cuda.set_device(0)
    _stream = torch.cuda.Stream()
eng = rsa.Engine(0)
if _stream is not None:
    eng.set_stream(_stream.cuda_stream)
eng.upload_table(nodes)
eng.set_model({"udgm": rsa.MODEL_UDGM, "logdist": rsa.MODEL_LOGDIST}[kind_name], **kw)
eng.set_link_capacity(1 << 21)
eng.events_enable(1 << 16, 1 << 21)
devs = [DeviceArray(W.choose_sources(n, t, 0xC0FFEE00 + idx, k)) for k in range(32)]


def loop(k0, k1):
    got = 0
    for k in range(k0, k1):
        eng.tick_run_sources_device(k * 1000, k * 1000 + 1000, devs[k % 32].ptr.value, t, k * 1000, W.AIR_US)
        got += len(eng.events_process(k * 1000 + 1000, copy=False, runs=True)[3])
    return got


loop(0, 24)
t0 = time.perf_counter()
got = loop(24, 24 + ticks)
dt = (time.perf_counter() - t0) / ticks
print(json.dumps({"workload": wl, "us_per_tick": dt * 1e6, "deliveries_per_tick": got / ticks, "links_per_s": t * (n - 1) / dt}))
eng.close()
