"""The C++ host-side mirror of the reference's plug-in API (radio-sim_amd/host/radiomedium.hpp:
Simulator / Node / Transciever / RadioPacket / *RadioMedium with the reference's names) driven like
a reference-side test, on the GPU; every Simulator call the medium makes is compared with what the
reference's loops would make (derived from the oracle): generateTransmissionEvents once, then
generateReceptionEvents(packet, node, rssi, doDeliver) per heard receiver in node order
(UDGMRadioMedium.java:97-111), deliverRadioPacket for the constant-loss medium
(UDGMConstantLossRadioMedium.java:31), event times per Simulator.java:323-333."""
import os
import subprocess

import numpy as np
import pytest

from util import KINDS

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")
HDR = os.path.join(ROOT, "radio-sim_amd", "host", "radiomedium.hpp")


def _build():
    lib = os.path.join(ROOT, "radio-sim_amd", "csrc")
    if (not os.path.exists(BIN)) or os.path.getmtime(BIN) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", BIN, SRC, "-L" + lib, "-lradiomedium_hip",
                               "-Wl,-rpath," + lib])
    return BIN


def _scenario(tmp_path, O, model, nd, ids, packets, extra, seed):
    lines = ["%s %d %d" % (model, seed, nd.n)]
    for i in range(nd.n):
        lines.append("%s %.17g %.17g %.17g %.17g %d %d %.17g %.17g" % (
            ids[i], nd.x[i], nd.y[i], nd.z[i], nd.txpower[i], nd.channel[i], nd.enabled[i], nd.rxprob[i], nd.txprob[i]))
    lines.append(extra)
    lines.append(str(len(packets)))
    for p in packets:
        if "move" in p:
            lines.append("@move %s %.17g %.17g %.17g" % ((ids[p["move"]],) + tuple(p["to"])))
            continue
        o = (" 1 %.17g %d" % (p["txpower"], p["channel"])) if "txpower" in p else " 0"
        lines.append("%s %d %d %s%s" % (p["id"], p["start"], p["now"], p["hex"] or "-", o))
    path = os.path.join(str(tmp_path), "scenario.txt")
    open(path, "w").write("\n".join(lines) + "\n")
    return path


def _run(path):
    out = subprocess.run([_build(), path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    calls, errors, name, base = [], [], None, None
    for ln in out.stdout.splitlines():
        f = ln.split()
        if f[0] == "call":
            calls.append((int(f[1]), int(f[2]), int(f[3]), float(f[4]), int(f[5]), int(f[6]), int(f[7])))
        elif f[0] == "error":
            errors.append(ln)
        elif f[0] == "name":
            name = ln[5:]
        elif f[0] == "base":
            base = (float(f[1]), float(f[2]))
    return name, base, calls, errors


@pytest.mark.parametrize("model", ["udgm", "const", "null", "n2n", "udgm@2", "n2n@3", "const@2"])
def test_mirror_makes_the_reference_calls(tmp_path, O, model):
    """`model@k`: the same medium as a GroupRadioMedium of k contexts on this GPU (rm_group_*: receivers split over the
    members, draw counts through the host) -- the calls must be those of the single context, i.e. the oracle's."""
    scenario_model = model
    model = model.split("@")[0]
    n = 400
    rng = np.random.default_rng(21)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    nd.x, nd.y = rng.uniform(0, side, n), rng.uniform(0, side, n)
    nd.channel[rng.random(n) < 0.1] = 25
    nd.enabled[rng.random(n) < 0.05] = 0
    nd.txpower[:] = rng.uniform(-20, 0, n)
    ids = [str(i + 1) for i in range(n)]
    ids[7] = "sensor-x"                      # non-numeric id -> getIdAsInteger() == -1
    nd.int_id[:] = np.arange(1, n + 1)
    nd.int_id[7] = -1
    seed = 4242
    okind, oparams, extra, matrix = {"udgm": ("udgm", {"udgm_success_ratio_rx": 0.7, "udgm_transmission_range": 60.0},
                                              "0.7 60.0", None),
                                     "const": ("udgm_const", {}, "", None),
                                     "null": ("null", {}, "", None),
                                     "n2n": ("n2n", {}, None, np.where(rng.random((n, n)) < 0.05, rng.uniform(0, 1.2, (n, n)), 0.0))}[model]
    if model == "udgm":
        nd.rxprob[:] = np.where(rng.random(n) < 0.5, 1.0, rng.uniform(0, 1, n))
    if model == "n2n":
        extra = "%d\n" % n + "\n".join(" ".join("%.17g" % v for v in row) for row in matrix)
    packets = []
    for k, s in enumerate(rng.choice(n, 25, replace=False)):
        p = {"id": ids[s], "src": int(s), "start": 1000 * k, "now": 1000 * k + (500 if k % 3 == 0 else 0),
             "hex": "0102030405" * (1 + k % 4)}
        if k % 5 == 0:
            p["txpower"], p["channel"] = -3.5, 26 if k % 10 else 25
        packets.append(p)
        if k % 4 == 1:     # node-config-set between packets: a neighbour of the next source moves next to it
            nxt = int(rng.integers(0, n))
            packets.append({"move": nxt, "to": (nd.x[s] + 3.0, nd.y[s] - 2.0, 0.5)})
    packets.append({"id": "nobody", "start": 0, "now": 0, "hex": "00"})       # unknown source
    packets.append({"id": ids[3], "src": 3, "start": 99000, "now": 0, "hex": ""})   # zero-length payload

    name, base, calls, errors = _run(_scenario(tmp_path, O, scenario_model, nd, ids, packets, extra, seed))
    assert base == (-100.0, -100.0)          # AbstractRadioMedium.java:38 ; Transciever.getRSSI() while idle
    assert name == {"udgm": "UDGM Radio Medium", "const": "UDGM Constant Loss Radio Medium",
                    "null": "Null radio medium - just forwards incoming packets to all other nodes",
                    "n2n": "Matrix Radio Medium"}[model]
    assert len(errors) == 1 and "could not find source node" in errors[0]

    kw = dict(oparams)
    if matrix is not None:
        kw["n2n_matrix"] = matrix
    mdl = O.model(KINDS[okind], **kw)
    state = O.lib().orc_jrandom_seed(seed)
    expected = []
    for k, p in enumerate(packets):
        if "move" in p:
            nd.x[p["move"]], nd.y[p["move"]], nd.z[p["move"]] = p["to"]
            continue
        if "src" not in p:
            continue
        pk = nd.packet(p["src"], p["start"], 32 * len(p["hex"]), txpower=p.get("txpower"), channel=p.get("channel"))
        r = O.tick(mdl, nd, pk, rng_state=state)
        state = r.rng_state
        t0 = max(p["start"], p["now"])
        t1 = t0 + 32 * len(p["hex"])
        if model != "const":
            expected.append((0, k, -1, 0.0, 0, t0, t1))
        for d, v, rssi in zip(r.dst, r.verdict, r.rssi):
            if model == "const":
                expected.append((2, k, int(d), float(rssi), 1, t0, t1))
            else:
                expected.append((1, k, int(d), float(rssi), int(v == O.DELIVERED), t0, t1))
    assert len(calls) == len(expected) > 100
    assert calls == expected
