"""The radio-link server end to end on the GPU: emulators and a time controller talk
JSON over TCP to radio-sim_amd/host/rsim_server, whose medium is the MI355X engine in tick mode with the events on
the device.  Every byte an emulator receives -- the time-step messages with their node-info, the receive messages
in the reference queue's pop order, the replies -- is compared with what the reference's server would send, built
here from the oracle's serial replay (media + EventQueue + Transciever state machine) and the reference's message
shapes (net/JSONClientConnection.java:326-388, RadioPacket.java:101-111)."""
import json
import os
import subprocess

import numpy as np
import pytest

from test_host_server import GREETING, Peer, java_double
from util import KINDS

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "radio-sim_amd", "host")
BIN = os.path.join(HOST, "rsim_server")


def _build():
    lib = os.path.join(ROOT, "radio-sim_amd", "csrc")
    deps = [os.path.join(HOST, f) for f in ("rsim_server.cpp", "json.hpp", "radiomedium.hpp")]
    if (not os.path.exists(BIN)) or os.path.getmtime(BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-o", BIN, deps[0], "-L" + lib, "-lradiomedium_hip", "-Wl,-rpath," + lib])
    return BIN


def num(x):
    s = java_double(float(x))
    return s[:-2] if s.endswith(".0") else s


def info_json(ident, rssi, state, channel):
    return '{"node-id":%s,"rssi":%s,"receiving":%d,"wireless-channel":%d}' % (json.dumps(ident), num(rssi), state, channel)


@pytest.mark.parametrize("model,mode", [("udgm", "tick"), ("udgm", "packet"), ("nullrm", "tick"), ("n2n-link", "tick"),
                                        ("log-distance", "tick"), ("log-distance", "packet"), ("log-distance-sinr", "tick"),
                                        ("udgm", "tick-workers"), ("nullrm", "tick-workers"), ("log-distance", "tick-workers")])
def test_server_end_to_end(O, model, mode):
    """`log-distance`: the engine's extension medium selected over the wire (the option string and its optional numeric
    parameters are the only additions to the protocol).  With "sinr": true the verdicts of the frames of one evaluation
    depend on each other, so the oracle evaluates what the server evaluates together: everything sent between two points
    where the server has to settle (a node-config-set, the end of the step), against the frames still on the air."""
    # "tick-workers": the node-info rewrites and the receive messages of every step go through the server's worker threads
    # (RSIM_PARALLEL_FROM=1: at the BASELINE sizes they do by themselves) -- every byte as before
    env = dict(os.environ)
    if mode == "tick-workers":
        env.update(RSIM_PARALLEL_FROM="1", RSIM_SERVER_THREADS="4")
        mode = "tick"
    sinr = model == "log-distance-sinr"
    n_reg = 90 if model != "n2n-link" else 24      # nodes registered before the first step
    late = model == "udgm"                         # one more node joins in mid-run (the node table grows under the medium)
    n = n_reg + (1 if late else 0)
    n_emu = 3
    rng = np.random.default_rng(91)
    nd = O.NodeTable(n)
    side = 50.0 * np.sqrt(np.pi * n / 12.0)
    nd.x, nd.y = np.round(rng.uniform(0, side, n), 3), np.round(rng.uniform(0, side, n), 3)
    nd.z[:] = np.where(rng.random(n) < 0.3, 1.5, 0.0)
    nd.channel[rng.random(n) < 0.1] = 25
    nd.enabled[rng.random(n) < 0.05] = 0
    nd.txpower[:] = np.round(rng.uniform(-20, 0, n), 2)
    nd.rxprob[:] = np.where(rng.random(n) < 0.5, 1.0, np.round(rng.uniform(0, 1, n), 3))
    nd.txprob[rng.random(n) < 0.2] = 0.6
    owner = rng.integers(0, n_emu, n)
    if late:
        nd.enabled[n - 1] = 0                      # until it joins, the oracle's table holds it as a radio that is off
        nd.rxprob[n - 1], nd.txprob[n - 1], nd.channel[n - 1] = 1.0, 1.0, 26
    seed = 4242
    if model == "n2n-link":
        matrix = np.round(rng.uniform(0, 1, (n, n)), 3)
        matrix[rng.random((n, n)) < 0.4] = 0.0
        mdl = O.model(KINDS["n2n"], n2n_matrix=matrix)
    elif model.startswith("log-distance"):
        mdl = O.model(KINDS["logdist"], ld_sigma_db=4.0, ld_seed=77, ld_exponent=3.2, ld_flags=1 if sinr else 0)
    else:
        mdl = O.model(KINDS["udgm" if model == "udgm" else "null"])

    args = [_build(), "--port", "0", "--bind", "127.0.0.1", "--seed", str(seed)] + (["--per-packet"] if mode == "packet" else [])
    proc = subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        first = proc.stdout.readline()
        assert first.startswith("Server started."), first + proc.stderr.read()
        port = int(first.rstrip(".\n").split()[-1])
        ctl = Peer(port)
        emus = [Peer(port) for _ in range(n_emu)]
        for p in [ctl] + emus:
            assert p.line() + b"\r\n" == GREETING
        params = {"wireless-standard": "802.15.4", "propagation-option": model}
        if model.startswith("log-distance"):
            params = {"propagation-option": "log-distance", "shadowing-sigma-db": 4.0, "shadowing-seed": 77, "path-loss-exponent": 3.2,
                      "sinr": sinr}
        if model == "n2n-link":
            params.update({"number-of-nodes": n, "matrix-data": [float(v) for v in matrix.reshape(-1)]})
        ctl.send({"command": "configuration-set", "id": 1, "parameters": params})
        assert ctl.line() == b'{"id":1,"reply":"OK"}'

        def sync(p, k=[100]):            # the server has handled everything this peer sent
            k[0] += 1
            p.send({"command": "time-get", "id": k[0]})
            assert p.line().startswith(b'{"id":%d,"reply":"OK"' % k[0])

        # node registration, in node order (the node table's order is the order of the receivers)
        for i in range(n_reg):
            prm = {"node-id": i + 1, "position": [float(nd.x[i]), float(nd.y[i]), float(nd.z[i])], "rf-power": float(nd.txpower[i]),
                   "wireless-channel": int(nd.channel[i]), "rx-loss": float(nd.rxprob[i]), "tx-loss": float(nd.txprob[i])}
            if not nd.enabled[i]:
                prm["radio-state"] = "disabled"
            emus[owner[i]].send({"command": "node-config-set", "id": i, "parameters": prm})
            want = '{"id":%d,"reply":"OK","reply-object":{"node-info":%s}}' % (i, info_json(str(i + 1), -100.0, 0 if nd.enabled[i] else 3,
                                                                                          nd.channel[i]))
            assert emus[owner[i]].line() == want.encode()

        state = O.lib().orc_jrandom_seed(seed)
        sim = O.Sim(n)
        pid, now = 0, 0
        n_rx = 0
        onair = np.zeros(0, dtype=O.PACKET_DTYPE)
        group = []          # (SINR) the packets of the evaluation the server has not made yet

        def close_group():
            """what the server evaluates in one go, against the frames on the air (rm_tick_begin's rule at the old time)"""
            nonlocal state, pid, onair
            if not group:
                return
            new = np.array(group, dtype=O.PACKET_DTYPE)
            onair = onair[onair["start_us"] + onair["air_us"] > now]
            r = O.tick(mdl, nd, np.concatenate([onair, new]), first_new=len(onair), rng_state=state)
            state = r.rng_state
            sim.medium_calls(r, new, pkt_base=pid)
            onair = np.concatenate([onair, new])
            pid += len(group)
            group.clear()
        hexes = ["", "0102030405", "0102030405" * 2, "0102030405" * 12, "ab" * 125]
        for t in range(16):
            step = now + int(rng.choice([1000, 1000, 1000, 10, 4000]))
            ctl.send({"command": "time-set", "id": 1000 + t, "parameters": {"time": step}})
            # every emulator is told, with the state its nodes had after the last drain
            rssi, st = sim.node_info(enabled=nd.enabled)
            for e in range(n_emu):
                mine = [i for i in range(n_reg) if owner[i] == e]
                want = '{"command":"time-step","id":%d,"parameters":{"time":%d,"node-info":[%s]}}' % (
                    1001 + t, step, ",".join(info_json(str(i + 1), rssi[i], st[i], nd.channel[i]) for i in mine))
                assert emus[e].line() == want.encode(), (t, e)
            imm = []
            for e in rng.permutation(n_emu):
                if late and t == 7 and e == owner[n - 1] and n_reg < n:      # a new node: the whole table is uploaded again
                    close_group()
                    i = n - 1
                    nd.enabled[i] = 1
                    emus[e].send({"command": "node-config-set", "id": 7000, "parameters": {
                        "node-id": i + 1, "position": [float(nd.x[i]), float(nd.y[i]), float(nd.z[i])], "rf-power": float(nd.txpower[i])}})
                    want = '{"id":7000,"reply":"OK","reply-object":{"node-info":%s}}' % info_json(str(i + 1), -100.0, 0, 26)
                    assert emus[e].line() == want.encode()
                    n_reg = n
                mine = [i for i in range(n_reg) if owner[i] == e]
                if t in (5, 9) and mine:     # a node moves / changes channel / loses its receiver in mid-run
                    close_group()                # (the server settles its queue before a node changes)
                    i = mine[int(rng.integers(len(mine)))]
                    nd.x[i], nd.y[i] = float(np.round(rng.uniform(0, side), 3)), float(np.round(rng.uniform(0, side), 3))
                    nd.channel[i] = 26
                    nd.rxprob[i] = 0.5
                    nd.enabled[i] = 0 if t == 9 else 1
                    emus[e].send({"command": "node-config-set", "parameters": {
                        "node-id": i + 1, "position": [float(nd.x[i]), float(nd.y[i])], "wireless-channel": 26, "rx-loss": 0.5,
                        "radio-state": "disabled" if t == 9 else "enabled"}})
                    nd.z[i] = 0.0            # Position.set(x, y) is set(x, y, 0.0)
                for s in rng.choice(mine, min(len(mine), int(rng.integers(0, 4))), replace=False):
                    hx = hexes[int(rng.integers(len(hexes)))]
                    start = int(rng.integers(now, step)) if t % 3 else now
                    msg = {"command": "transmit", "node-id": int(s) + 1, "time": start, "packet-data": hx}
                    txp = ch = None
                    if rng.random() < 0.25:
                        txp, ch = -3.5, 26
                        msg["rf-power"], msg["wireless-channel"] = txp, ch
                    emus[e].send(msg)
                    rec = nd.packet(int(s), start, 32 * len(hx), txpower=txp, channel=ch)
                    if sinr:
                        packets_hex[pid + len(group)] = (hx, start, ch if ch is not None else int(nd.channel[s]))
                        group.append(rec)
                        continue
                    r = O.tick(mdl, nd, rec, rng_state=state)
                    state = r.rng_state
                    sim.medium_calls(r, np.atleast_1d(rec), pkt_base=pid)
                    packets_hex[pid] = (hx, start, ch if ch is not None else int(nd.channel[s]))
                    pid += 1
                sync(emus[e])
            close_group()
            for e in range(n_emu):
                emus[e].send({"reply": "OK", "id": 1001 + t})
            # the drain: receive messages per destination's connection, in pop order, then the controller's reply
            ev = sim.step(step)
            per_conn = [[] for _ in range(n_emu)]
            for evn in ev:
                if evn["kind"] == O.EV_RX_END_DELIVERY:
                    hx, start, ch = packets_hex[int(evn["pkt"])]
                    d = int(evn["node"])
                    per_conn[owner[d]].append('{"command":"receive","node-id":"%d","time-start":%d,"time-end":%d,"rf-power":%s,'
                                              '"wireless-channel":%d,"packet-data":"%s"}' % (d + 1, start, start + 32 * len(hx),
                                                                                             num(evn["rssi"]), ch, hx))
            assert ctl.line() == b'{"reply":"OK","id":%d}' % (1000 + t)
            for e in range(n_emu):
                for want in per_conn[e]:
                    assert emus[e].line() == want.encode(), (t, e)
                    n_rx += 1
            now = step
        assert pid > 20 and n_rx > 30, (pid, n_rx)
        for p in [ctl] + emus:
            p.close()
    finally:
        proc.terminate()
        try:
            proc.wait(timeout=10)
        except subprocess.TimeoutExpired:
            proc.kill()
        err = proc.stderr.read()
    assert "radio medium error" not in err, err


packets_hex = {}


def test_server_keeps_its_packets_straight_when_an_evaluation_fails():
    """a node-config-set with a position of [1e999, 0] (Double.parseDouble gives Infinity) used to make every
    later evaluation fail, and the server then freed OLDER packets whose end events were still pending.  Now the position
    is not applied (stderr says so), nothing fails, and long frames sent before and after it are delivered to the end."""
    n = 12
    args = [_build(), "--port", "0", "--bind", "127.0.0.1", "--seed", "7"]
    proc = subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        first = proc.stdout.readline()
        port = int(first.rstrip(".\n").split()[-1])
        ctl, emu = Peer(port), Peer(port)
        for p in (ctl, emu):
            assert p.line() + b"\r\n" == GREETING
        ctl.send({"command": "configuration-set", "id": 1, "parameters": {"propagation-option": "udgm"}})
        assert ctl.line() == b'{"id":1,"reply":"OK"}'
        for i in range(n):
            emu.send({"command": "node-config-set", "id": i, "parameters": {"node-id": i + 1, "position": [float(3 * i), 0.0, 0.0]}})
            assert emu.line().startswith(b'{"id":%d,"reply":"OK"' % i)
        long_hex = "ab" * 127            # 8128 us on the air: pending over eight 1 ms steps
        received, now = 0, 0
        for t in range(14):
            step = now + 1000
            ctl.send({"command": "time-set", "id": 1000 + t, "parameters": {"time": step}})
            assert emu.line().startswith(b'{"command":"time-step"')
            emu.send({"command": "transmit", "node-id": 1 + (t % n), "time": now, "packet-data": long_hex})
            if t == 2:
                emu.raw('{"command":"node-config-set","id":500,"parameters":{"node-id":4,"position":[1e999,0]}}')
                assert emu.line().startswith(b'{"id":500,"reply":"OK"')
            emu.send({"command": "transmit", "node-id": 1 + ((t + 5) % n), "time": now, "packet-data": "0102"})
            emu.send({"reply": "OK", "id": 1001 + t})
            # receive messages of this drain, then the controller's OK
            assert ctl.line() == b'{"reply":"OK","id":%d}' % (1000 + t)
            emu.send({"command": "time-get", "id": 9000 + t})
            while True:
                ln = emu.line()
                if ln.startswith(b'{"id":%d,' % (9000 + t)):
                    break
                assert ln.startswith(b'{"command":"receive"'), ln
                body = json.loads(ln)
                assert body["packet-data"] in (long_hex, "0102")
                received += 1
            now = step
        # every frame is heard by the 11 other nodes (range 50, nodes 3 m apart); the long frames of the first six steps
        # have ended by now
        assert received >= 14 * 11 + 6 * 11, received
        assert proc.poll() is None
        for p in (ctl, emu):
            p.close()
    finally:
        proc.terminate()
        try:
            proc.wait(timeout=10)
        except subprocess.TimeoutExpired:
            proc.kill()
        err = proc.stderr.read()
    assert "non-finite position ignored" in err
    assert "radio medium error" not in err, err
