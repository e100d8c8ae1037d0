"""Closed-loop tick latency: one simulated tick at a time, the reference's call pattern (the emulators
step, their transmit() calls are evaluated, the events are consumed in emulatorTimeStepDone,
Simulator.java:155-165, and only then does the next tick begin).

  device-resident     rm_tick_run_sources_device, results left in HBM (what bench.py calls sequential_ticks)
  host buffers        rm_tick_begin / rm_enqueue_tx_records / rm_tick_flush_view: records in over PCIe, heard links
                      read in place from the host-mapped block (what a JNI tick mode sees)

Run on the GPU box:  python tools/tick_latency.py [workload] [ticks]      (RM_FRAME_TICK=0: the three-launch sweep)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import radio_sim_amd as rsa  # noqa: E402
from radio_sim_amd import workload as W  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import DeviceArray  # noqa: E402

SHAPES = {"c2": (2, 10_000, 0.01, "logdist"), "c3": (3, 100_000, 0.01, "logdist_shadow"), "udgm": (3, 100_000, 0.01, "udgm"),
          "udgm_lossy": (3, 100_000, 0.01, "udgm_lossy"), "m1": (5, 1_000_000, 0.001, "logdist_shadow")}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    idx, n, frac, model = SHAPES[wl]
    t = int(round(n * frac))
    nodes = W.make_nodes(n, idx)
    kind_name, kw = W.model_kwargs(model)
    kind = {"udgm": rsa.MODEL_UDGM, "udgm_const": rsa.MODEL_UDGM_CONST, "logdist": rsa.MODEL_LOGDIST}[kind_name]
    eng = rsa.Engine(0)
    eng.upload_table(nodes)
    eng.set_model(kind, **kw)
    eng.set_link_capacity(1 << 21)
    eng.seed(1)
    pool = 64
    srcs = [W.choose_sources(n, t, 0xC0FFEE00 + idx, k) for k in range(pool)]
    dev = [DeviceArray(s) for s in srcs]
    out = {"workload": wl, "nodes": n, "frames_per_tick": t, "frame_tick": os.environ.get("RM_FRAME_TICK", "default")}

    def device_loop(k0, k1):
        for k in range(k0, k1):
            eng.tick_run_sources_device(k * W.TICK_US, (k + 1) * W.TICK_US, dev[k % pool].ptr.value, t, k * W.TICK_US, W.AIR_US)
        eng.sync()

    device_loop(0, 50)
    t0 = time.perf_counter()
    device_loop(50, 50 + ticks)
    dt = (time.perf_counter() - t0) / ticks
    heard, dropped = eng.result_count()
    out["device_resident_us_per_tick"] = dt * 1e6
    out["device_resident_links_per_s"] = t * (n - 1) / dt
    out["heard_last_tick"] = int(heard)
    assert not dropped

    recs = []
    for s in srcs:
        r = np.zeros(t, dtype=rsa.TX_RECORD_DTYPE)
        r["x"], r["y"], r["z"] = nodes.x[s], nodes.y[s], nodes.z[s]
        r["txpower"], r["txprob"], r["channel"] = nodes.txpower[s], nodes.txprob[s], nodes.channel[s]
        r["src"], r["air_us"] = s, W.AIR_US
        recs.append(r)

    def host_loop(k0, k1):
        links = 0
        for k in range(k0, k1):
            eng.tick_begin(k * W.TICK_US, (k + 1) * W.TICK_US)
            eng.enqueue_records(recs[k % pool])
            links += eng.tick_flush_view().count
        return links

    host_loop(0, 20)
    reps = max(20, ticks // 4)
    t0 = time.perf_counter()
    links = host_loop(20, 20 + reps)
    dt = (time.perf_counter() - t0) / reps
    out["host_buffers_us_per_tick"] = dt * 1e6
    out["host_buffers_links_per_s"] = t * (n - 1) / dt
    out["host_buffers_bytes_out_per_tick"] = links / reps * 25
    print(json.dumps(out))
    for d in dev:
        d.free()
    eng.close()


if __name__ == "__main__":
    main()
