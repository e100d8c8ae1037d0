"""A minimal emulator + time controller against the radio-link server (radio-sim_amd/host/rsim_server) -- or against the
reference's own server: the protocol is the same.

    radio-sim_amd/host/rsim_server --port 7711 &
    python examples/rsim_client.py [host] [port]

Registers three nodes 30 m apart, selects the UDGM medium, steps the time by 1 ms five times and lets node 1 transmit
in every step; prints what the server sends back (time-step with node-info, receive, replies)."""
import json
import socket
import sys

host = sys.argv[1] if len(sys.argv) > 1 else "127.0.0.1"
port = int(sys.argv[2]) if len(sys.argv) > 2 else 7711
s = socket.create_connection((host, port))
s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
f = s.makefile("rb")


def send(obj):
    s.sendall(json.dumps(obj, separators=(",", ":")).encode() + b"\r\n")


def recv():
    return json.loads(f.readline())


print("greeting:", recv())
send({"command": "configuration-set", "id": 1, "parameters": {"wireless-standard": "802.15.4", "propagation-option": "udgm"}})
print(recv())
for i, x in enumerate((0.0, 30.0, 60.0)):
    send({"command": "node-config-set", "id": 10 + i, "parameters": {"node-id": i + 1, "position": [x, 0.0, 0.0]}})
    print(recv())
for step in range(1, 6):
    send({"command": "time-set", "id": 100 + step, "parameters": {"time": step * 1000}})   # this connection is the time controller ...
    while True:                                                                            # ... and the emulator of the three nodes
        m = recv()
        print(m)
        if m.get("command") == "time-step":
            send({"command": "transmit", "node-id": 1, "time": (step - 1) * 1000 + 100, "packet-data": "61dc%02x0102030405" % step})
            send({"reply": "OK", "id": m["id"]})
        elif m.get("reply") == "OK" and m.get("id") == 100 + step:
            break
