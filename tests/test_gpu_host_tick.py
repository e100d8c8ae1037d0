"""The C++ mirror of the reference's plug-in API in its batching modes (radio-sim_amd/host/radiomedium.hpp), driven
tick by tick on the GPU through the C ABI:

  packet   one rm_transmit per RadioMedium.transmit (the mode tests/test_gpu_host_mirror.py covers call by call)
  tick     transmit() queues; Simulator::emulatorTimeStepDone flushes the queue in ONE evaluation before the time
           moves (the reference consumes a tick's events only there, Simulator.java:155-165) -- the medium must
           make exactly the per-packet mode's Simulator calls, in the same order
  device   the events stay on the device (rm_events_*): the drain's deliverRadioPacket calls and the node-info of
           every node must equal the oracle's serial replay of the reference's queue and state machine"""
import os
import subprocess

import numpy as np
import pytest

from util import KINDS

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_tick_test.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "host_tick_test")
HDR = os.path.join(ROOT, "radio-sim_amd", "host", "radiomedium.hpp")


def _build():
    lib = os.path.join(ROOT, "radio-sim_amd", "csrc")
    if (not os.path.exists(BIN)) or os.path.getmtime(BIN) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", BIN, SRC, "-L" + lib, "-lradiomedium_hip",
                               "-Wl,-rpath," + lib])
    return BIN


def _run(path, mode):
    out = subprocess.run([_build(), path, mode], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    calls, info, errors = [], {}, []
    for ln in out.stdout.splitlines():
        f = ln.split()
        if f[0] == "call":
            calls.append((int(f[1]), int(f[2]), int(f[3]), float(f[4]), int(f[5]), int(f[6]), int(f[7])))
        elif f[0] == "info":
            info.setdefault(int(f[1]), {})[int(f[2])] = (float(f[3]), int(f[4]), int(f[5]))
        elif f[0] == "error":
            errors.append(ln)
    return calls, info, errors


@pytest.mark.parametrize("model", ["udgm", "const", "null"])
def test_tick_mode_and_device_events(tmp_path, O, model):
    n = 700 if model != "null" else 120
    rng = np.random.default_rng(33)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.channel[rng.random(n) < 0.1] = 25
    nd.enabled[rng.random(n) < 0.05] = 0
    nd.txpower[:] = rng.uniform(-20, 0, n)
    if model == "udgm":
        nd.rxprob[:] = np.where(rng.random(n) < 0.5, 1.0, rng.uniform(0, 1, n))
        nd.txprob[rng.random(n) < 0.2] = 0.6
    ids = [str(i + 1) for i in range(n)]
    seed = 777
    okind, oparams, extra = {"udgm": ("udgm", {"udgm_success_ratio_rx": 0.7, "udgm_transmission_range": 60.0}, "0.7 60.0"),
                             "const": ("udgm_const", {}, ""), "null": ("null", {}, "")}[model]
    lines = ["%s %d %d" % (model, seed, n)]
    for i in range(n):
        lines.append("%s %.17g %.17g %.17g %.17g %d %d %.17g %.17g" % (
            ids[i], nd.x[i], nd.y[i], nd.z[i], nd.txpower[i], nd.channel[i], nd.enabled[i], nd.rxprob[i], nd.txprob[i]))
    lines.append(extra)
    ticks, now = [], 0
    for t in range(14):
        step = now + int(rng.choice([1000, 1000, 1000, 10, 4000]))
        pk = []
        for s in rng.choice(n, int(rng.integers(0, 9)), replace=False):
            p = {"src": int(s), "start": int(rng.integers(now, step)) if t % 3 else now,
                 "hex": "0102030405" * int(rng.choice([0, 1, 2, 12, 25]))}
            if rng.random() < 0.2:
                p["txpower"], p["channel"] = -3.5, 26
            pk.append(p)
        ticks.append((step, pk))
        now = step
    ticks.append((now + 100000, []))       # a last drain
    lines.append(str(len(ticks)))
    for step, pk in ticks:
        lines.append("%d %d" % (step, len(pk)))
        for p in pk:
            o = (" 1 %.17g %d" % (p["txpower"], p["channel"])) if "txpower" in p else " 0"
            lines.append("%s %d %s%s" % (ids[p["src"]], p["start"], p["hex"] or "-", o))
    path = os.path.join(str(tmp_path), "ticks.txt")
    open(path, "w").write("\n".join(lines) + "\n")

    per_packet, _, e1 = _run(path, "packet")
    per_tick, _, e2 = _run(path, "tick")
    device, info, e3 = _run(path, "device")
    assert not (e1 or e2 or e3), (e1, e2, e3)

    # what the reference's loops call, from the oracle (one java.util.Random through all packets)
    mdl = O.model(KINDS[okind], **oparams)
    state = O.lib().orc_jrandom_seed(seed)
    sim = O.Sim(n)
    expected, deliveries, want_info = [], [], {}
    pid, cur = 0, 0
    for t, (step, pk) in enumerate(ticks):
        imm = []
        for p in pk:
            rec = nd.packet(p["src"], p["start"], 32 * len(p["hex"]), txpower=p.get("txpower"), channel=p.get("channel"))
            r = O.tick(mdl, nd, rec, rng_state=state)
            state = r.rng_state
            t0 = max(p["start"], cur)
            t1 = t0 + 32 * len(p["hex"])
            if model != "const":
                expected.append((0, pid, -1, 0.0, 0, t0, t1))
            for d, v, rssi in zip(r.dst, r.verdict, r.rssi):
                expected.append((2, pid, int(d), float(rssi), 1, t0, t1) if model == "const" else
                                (1, pid, int(d), float(rssi), int(v == O.DELIVERED), t0, t1))
            imm += sim.medium_calls(r, np.atleast_1d(rec), pkt_base=pid, const_loss=(model == "const"))
            pid += 1
        ev = sim.step(step)
        deliveries += [(q, d, r) for q, d, r in imm]
        deliveries += [(int(e["pkt"]), int(e["node"]), float(e["rssi"])) for e in ev if e["kind"] == O.EV_RX_END_DELIVERY]
        rssi, st = sim.node_info(enabled=nd.enabled)
        want_info[t] = {i: (float(rssi[i]), int(st[i]), int(nd.channel[i])) for i in range(n) if st[i] != 0 or rssi[i] != -100.0}
        cur = step
    assert len(expected) > 60
    assert per_packet == expected
    assert per_tick == expected                      # ONE evaluation per tick, the same calls in the same order
    # device events: only the deliveries reach the host, in the reference queue's pop order
    assert [c[0] for c in device] == [2] * len(device)
    assert [(c[1], c[2], c[3]) for c in device] == deliveries and len(deliveries) > 20
    assert info == {t: v for t, v in want_info.items() if v}
