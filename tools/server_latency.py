"""Round trip of one simulated tick through the radio-link server (radio-sim_amd/host/rsim_server): a time controller and
one emulator connection over TCP on the GPU box, UDGM medium, `n` nodes, `k` transmissions per tick.
    python tools/server_latency.py [n] [k] [ticks] [packet bytes]
(packet bytes: 8 by default -- a frame shorter than the tick, every radio idle again when the step ends; 127 is a full
802.15.4 frame, 4 ms on the air: receptions span the steps, and thousands of nodes report a changed state every step)
What is timed is what an emulator sees: time-set -> time-step (with node-info of all its nodes) -> k transmit commands
-> its OK -> the receive messages -> the controller's reply.  The Python client's JSON work is part of it."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from test_host_server import GREETING, Peer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 200
pkt_hex = "".join("%02x" % (i & 255) for i in range(int(sys.argv[4]) if len(sys.argv) > 4 else 8))
exe = os.path.join(ROOT, "radio-sim_amd", "host", "rsim_server")
proc = subprocess.Popen([exe, "--port", "0", "--bind", "127.0.0.1", "--seed", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
try:
    port = int(proc.stdout.readline().rstrip(".\n").split()[-1])
    ctl, emu = Peer(port), Peer(port)
    assert ctl.line() + b"\r\n" == GREETING and emu.line() + b"\r\n" == GREETING
    ctl.send({"command": "configuration-set", "id": 1, "parameters": {"propagation-option": "udgm"}})
    ctl.line()
    rng = np.random.default_rng(3)
    side = 50.0 * np.sqrt(np.pi * n / 20.0)
    for i in range(n):
        emu.send({"command": "node-config-set", "parameters": {"node-id": i + 1, "position": [float(rng.uniform(0, side)), float(rng.uniform(0, side))]}})
    emu.send({"command": "time-get", "id": 2})
    emu.line()
    got = 0

    def tick(t):
        global got
        ctl.send({"command": "time-set", "id": 100 + t, "parameters": {"time": (t + 1) * 1000}})
        while True:   # the receive messages of the step before come first
            step = json.loads(emu.line())
            if step.get("command") == "time-step":
                break
            got += 1
        for s in rng.choice(n, k, replace=False):
            emu.send({"command": "transmit", "node-id": int(s) + 1, "time": t * 1000 + 10, "packet-data": pkt_hex})
        emu.send({"reply": "OK", "id": step["id"]})
        ctl.line()
        return step

    for t in range(20):
        tick(t)
    t0 = time.perf_counter()
    for t in range(20, 20 + ticks):
        tick(t)
    dt = (time.perf_counter() - t0) / ticks
    emu.send({"command": "time-get", "id": 3})
    while True:
        m = json.loads(emu.line())
        if m.get("command") == "receive":
            got += 1
        if m.get("id") == 3 and "reply" in m:
            break
    print(json.dumps({"nodes": n, "tx_per_tick": k, "ticks": ticks, "packet_bytes": len(pkt_hex) // 2, "us_per_tick": dt * 1e6, "receive_messages": got}))
finally:
    proc.terminate()
    proc.wait(timeout=10)
    print(proc.stderr.read().strip().splitlines()[-1])   # the server's own share per step
