"""Where the SINR medium's tick by scan spends its time INSIDE k_sinr_scan (rm_airscan.hip): in-kernel s_memtime stamps of
the diagnostic build (make -C radio-sim_amd/csrc stamps).  Shares, not durations: the stamped build is not the shipped one.
    RM_LIBRARY=radio-sim_amd/csrc/libradiomedium_hip_stamps.so python tools/scan_stamps.py [ticks]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import radio_sim_amd as rsa  # noqa: E402
from radio_sim_amd import workload as W  # noqa: E402
from util import DeviceArray  # noqa: E402

idx, n, frac, model = 5, 1_000_000, 0.001, "logdist_sinr_overlap"
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t = int(round(n * frac))
nodes = W.make_nodes(n, idx)
kind_name, kw = W.model_kwargs(model)
eng = rsa.Engine(0)
eng.upload_table(nodes)
eng.set_model(rsa.MODEL_LOGDIST, **kw)
cap = 1 << 25
eng.set_link_capacity(cap)
devs = [DeviceArray(W.choose_sources(n, t, 0xC0FFEE00 + idx, k)) for k in range(ticks)]
for k in range(ticks):
    eng.tick_run_sources_device(k * 1000, k * 1000 + 1000, devs[k].ptr.value, t, k * 1000, W.AIR_US)
eng.sync()
res = eng.result_device()          # (compaction of the last tick: writes the head of the arrays only)
eng.sync()
raw = DeviceArray.read(res.rssi + 8 * (cap - 32 * 1024), np.uint64, 16 * 1024).reshape(1024, 16)[::-1][:t]
raw = raw[raw[:, 13] != 0]          # (frames nobody heard return before the first stamp)
st = raw[:, :7].astype(np.int64)
names = ["near list", "links", "near records", "pair tests", "exact", "verdicts"]
d = np.diff(st, axis=1)
wall = (raw[:, 14].astype(np.int64) - raw[:, 15].astype(np.int64)) * 10.0   # 100 MHz ticks -> ns
print("k_sinr_scan after %d ticks: per-frame workgroup, cycles between stamps (median / p90), %d frames" % (ticks, len(st)))
for i, nm in enumerate(names):
    print("  %-12s %7.0f %7.0f" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90)))
print("  near frames per new frame: median %d, max %d ; heard links: median %d, max %d" % (
    np.median(raw[:, 13] & 0xFFFFFFFF), (raw[:, 13] & 0xFFFFFFFF).max(), np.median(raw[:, 13] >> 32), (raw[:, 13] >> 32).max()))
print("  grid cells looked at per new frame: median %d (0: every frame on the air instead)" % np.median(raw[:, 12]))
print("  total cycles %7.0f ; wall per workgroup %.2f us (median), first start -> last end %.2f us" % (
    np.median(st[:, 6] - st[:, 0]), np.median(wall) / 1e3,
    (raw[:, 14].astype(np.int64).max() - raw[:, 15].astype(np.int64).min()) * 10.0 / 1e3))
print("  start skew: last workgroup starts %.2f us after the first" % ((raw[:, 15].astype(np.int64).max() - raw[:, 15].astype(np.int64).min()) * 10.0 / 1e3))
