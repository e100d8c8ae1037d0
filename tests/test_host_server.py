"""CPU tier: the radio-link server's protocol layer (radio-sim_amd/host/rsim_server.cpp,
json.hpp) against the message shapes of the reference (net/SimulatorJSONHandler.java,
net/JSONClientConnection.java, net/Server.java, Simulator.java -- line numbers in the server's header).  The server
runs with --no-medium: no device, no evaluation, "transmit" answers as the reference does without a medium.
The end-to-end test with the MI355X medium behind the same sockets is tests/test_gpu_server.py."""
import json
import os
import socket
import struct
import subprocess
import time
from decimal import Decimal

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "radio-sim_amd", "host")
GREETING = b'{"radio-simulator":{"name":"RSIM 0.1","api-version":"0.6"},"status":"OK"}\r\n'


def build_server(rsa, tmp_path):
    if os.environ.get("RSIM_SERVER_EXE"):      # e.g. a sanitizer build of the same source
        return os.environ["RSIM_SERVER_EXE"]
    lib = os.path.dirname(rsa.library_path())
    exe = os.path.join(str(tmp_path), "rsim_server")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-o", exe, os.path.join(HOST, "rsim_server.cpp"),
                           "-L" + lib, "-lradiomedium_hip", "-Wl,-rpath," + lib])
    return exe


class Peer:
    def __init__(self, port):
        self.s = socket.create_connection(("127.0.0.1", port), timeout=10)
        self.s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)   # many small writes: no 40 ms Nagle / delayed-ACK stalls
        self.buf = b""

    def raw(self, data):
        self.s.sendall(data if isinstance(data, bytes) else data.encode())

    def send(self, obj):
        self.raw(json.dumps(obj, separators=(",", ":")))

    def line(self, timeout=10.0):
        """one CR LF terminated message, as bytes without the terminator"""
        self.s.settimeout(timeout)
        while b"\r\n" not in self.buf:
            chunk = self.s.recv(65536)
            if not chunk:
                raise EOFError("closed")
            self.buf += chunk
        out, self.buf = self.buf.split(b"\r\n", 1)
        return out

    def closed(self, timeout=5.0):
        self.s.settimeout(timeout)
        try:
            while True:
                chunk = self.s.recv(65536)
                if not chunk:
                    return True
                self.buf += chunk
        except socket.timeout:
            return False
        except ConnectionError:
            return True

    def close(self):
        self.s.close()


@pytest.fixture(scope="module")
def server_exe(rsa, tmp_path_factory):
    return build_server(rsa, tmp_path_factory.mktemp("rsim"))


@pytest.fixture
def server(server_exe):
    p = subprocess.Popen([server_exe, "--no-medium", "--port", "0", "--bind", "127.0.0.1"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    first = p.stdout.readline()
    assert first.startswith("Server started. Waiting for client connections at port "), first
    port = int(first.rstrip(".\n").split()[-1])
    yield port
    p.terminate()
    try:
        p.wait(timeout=5)
    except subprocess.TimeoutExpired:
        p.kill()


def connect(port):
    c = Peer(port)
    assert c.line() + b"\r\n" == GREETING           # Server.java:60-66,107-109
    return c


def test_greeting_and_simple_commands(server):
    c = connect(server)
    c.send({"command": "time-get", "id": 1})
    assert c.line() == b'{"id":1,"reply":"OK","reply-object":{"time":0}}'
    c.send({"command": "time-get"})                   # no id: no reply
    c.send({"id": 2})
    assert c.line() == b'{"id":2,"reply":"error","reply-object":{"class":"command-error","description":"no command specified"}}'
    c.send({"nothing": 1})
    assert c.line() == b'{"reply":"error","reply-object":{"class":"command-error","description":"no command specified"}}'
    c.send({"command": "fly", "id": 3})
    assert c.line() == b'{"id":3,"reply":"error","reply-object":{"class":"command-error","description":"unsupported command: fly"}}'
    c.send({"command": "subscribe-event", "id": 4})
    assert c.line() == b'{"id":4,"reply":"OK"}'
    c.send({"command": "link-quality", "id": 5, "link": {"src": 1, "dst": 2, "quality": 90}, "wireless-channel": 26})
    assert c.line() == b'{"id":5,"reply":"OK"}'
    c.send({"command": "time-set", "parameters": {"time": 5}})
    assert c.line() == b'{"reply":"error","reply-object":{"class":"command-error","description":"time-set must include reply id"}}'
    c.close()


def test_time_controller_rules(server):
    a, b = connect(server), connect(server)
    a.send({"command": "configuration-set", "id": 1, "parameters": {"wireless-standard": "802.15.4", "propagation-option": "nullrm"}})
    assert a.line() == b'{"id":1,"reply":"OK"}'
    a.send({"command": "time-set", "id": 2, "parameters": {"time": 1000}})
    assert a.line() == b'{"reply":"OK","id":2}'        # timeStepDone: reply first, then id (JSONClientConnection.java:372)
    a.send({"command": "time-get", "id": 3})
    assert a.line() == b'{"id":3,"reply":"OK","reply-object":{"time":1000}}'
    b.send({"command": "time-set", "id": 4, "parameters": {"time": 2000}})
    assert b.line() == b'{"id":4,"reply":"error","reply-object":{"class":"command-error","description":"only one time controller allowed"}}'
    b.send({"command": "configuration-set", "id": 5, "parameters": {"propagation-option": "udgm"}})
    assert b.line() == b'{"id":5,"reply":"error","reply-object":{"class":"command-error","description":"already initialized"}}'
    a.raw('{"command":"time-set","id":6,"parameters":{"time":1.5}}')
    assert a.line() == (b'{"id":6,"reply":"error","reply-object":{"class":"command-error","description":'
                        b'"failed to set time:For input string: \\"1.5\\""}}')
    a.raw('{"command":"time-set","id":7,"parameters":5}')
    assert a.line() == b'{"id":7,"reply":"error","reply-object":{"class":"command-error","description":"failed to set time:Not an object: 5"}}'
    a.close()
    b.close()


def test_configuration_set_n2n_checks(server):
    c = connect(server)
    c.send({"command": "configuration-set", "id": 1, "parameters": {"propagation-option": "n2n-link", "number-of-nodes": 2}})
    assert c.line() == b'{"id":1,"reply":"error","reply-object":{"class":"command-error","description":"no matrix specified"}}'
    c.send({"command": "configuration-set", "id": 2, "parameters": {"propagation-option": "n2n-link", "number-of-nodes": 3,
                                                                    "matrix-data": [1, 0.5, 0.5, 1]}})
    assert c.line() == b'{"id":2,"reply":"error","reply-object":{"class":"command-error","description":"inconsistent data matrix or nodes"}}'
    c.send({"command": "configuration-set", "id": 3, "parameters": {"propagation-option": "quantum"}})
    assert c.line() == b'{"id":3,"reply":"OK"}'         # unsupported option: logged, the null medium stays
    # the two option strings this server adds (the engine's extension medium with its optional parameters; the constant-
    # loss medium, which the reference has as a class but never wired to the protocol)
    c.send({"command": "configuration-set", "id": 4, "parameters": {"propagation-option": "log-distance", "path-loss-exponent": 3.0,
                                                                    "shadowing-sigma-db": 4.0, "shadowing-seed": 7, "sinr": True}})
    assert c.line() == b'{"id":4,"reply":"OK"}'
    c.send({"command": "configuration-set", "id": 5, "parameters": {"propagation-option": "udgm-constant-loss"}})
    assert c.line() == b'{"id":5,"reply":"OK"}'
    c.close()


def test_nodes_time_steps_and_events(server):
    ctl, emu, emu2 = connect(server), connect(server), connect(server)
    emu.send({"command": "node-config-set", "id": 1, "parameters": {"node-id": 1, "position": [1.0, 2.0], "rf-power": -3.5,
                                                                      "wireless-channel": 11}})
    assert emu.line() == (b'{"id":1,"reply":"OK","reply-object":{"node-info":{"node-id":"1","rssi":-99.99,"receiving":0,'
                          b'"wireless-channel":11}}}')
    emu.raw('{"command":"node-config-set","id":2,"parameters":{"node-id":"n2","radio-state":"disabled"}}')
    assert emu.line() == (b'{"id":2,"reply":"OK","reply-object":{"node-info":{"node-id":"\\"n2\\"","rssi":-99.99,"receiving":3,'
                          b'"wireless-channel":26}}}')
    emu2.send({"command": "node-config-set", "parameters": {"node-id": 7, "position": [0, 0, 1]}})   # no id: no reply
    emu2.send({"command": "node-config-set", "id": 3, "parameters": {"node-id": 1, "wireless-channel": 12}})  # an existing node keeps its owner
    assert emu2.line() == (b'{"id":3,"reply":"OK","reply-object":{"node-info":{"node-id":"1","rssi":-99.99,"receiving":0,'
                           b'"wireless-channel":12}}}')
    # transmit: unknown node, then no medium (this server runs without one)
    emu.send({"command": "transmit", "id": 4, "node-id": 99, "time": 0, "packet-data": "00"})
    assert emu.line() == b'{"id":4,"reply":"error","reply-object":{"class":"command-error","description":"could not find source node"}}'
    emu.send({"command": "transmit", "id": 5, "node-id": 1, "time": 0, "packet-data": "00"})
    assert emu.line() == b'{"id":5,"reply":"error","reply-object":{"class":"command-error","description":"no radio medium available"}}'
    # a time step: both emulators are told, with their own nodes' info, ids count from 1001 (Simulator.java:78,118)
    ctl.send({"command": "time-set", "id": 10, "parameters": {"time": 1000}})
    assert emu.line() == (b'{"command":"time-step","id":1001,"parameters":{"time":1000,"node-info":['
                          b'{"node-id":"1","rssi":-99.99,"receiving":0,"wireless-channel":12},'
                          b'{"node-id":"\\"n2\\"","rssi":-99.99,"receiving":3,"wireless-channel":26}]}}')
    assert emu2.line() == (b'{"command":"time-step","id":1001,"parameters":{"time":1000,"node-info":['
                           b'{"node-id":"7","rssi":-99.99,"receiving":0,"wireless-channel":26}]}}')
    ctl.send({"command": "time-get", "id": 11})
    assert ctl.line() == b'{"id":11,"reply":"OK","reply-object":{"time":0}}'   # not there yet
    emu.send({"reply": "OK", "id": 1000})               # not the id waited for: ignored
    emu.send({"reply": "OK", "id": 1001})
    emu2.send({"reply": "error", "id": 1001})           # an error reply is logged only
    emu2.send({"reply": "OK", "id": 1001})
    assert ctl.line() == b'{"reply":"OK","id":10}'
    ctl.send({"command": "time-get", "id": 12})
    assert ctl.line() == b'{"id":12,"reply":"OK","reply-object":{"time":1000}}'
    # log events go to the subscribers, in subscription order, with the simulator's time
    ctl.send({"command": "subscribe-event"})
    emu2.send({"command": "subscribe-event", "id": 13})
    assert emu2.line() == b'{"id":13,"reply":"OK"}'
    emu.send({"command": "log", "id": 14, "parameters": {"node-id": 1, "message": "hello \"radio\""}})
    assert emu.line() == b'{"id":14,"reply":"OK"}'
    want = b'{"event":{"time":1000,"type":"log","source":"1","event-data":{"logMessage":"hello \\"radio\\""}},"id":0}'
    assert ctl.line() == want and emu2.line() == want
    ctl.send({"command": "unsubscribe-event", "id": 15})
    assert ctl.line() == b'{"id":15,"reply":"OK"}'
    emu.send({"command": "log", "parameters": {"node-id": 1, "message": "again"}})
    assert emu2.line() == b'{"event":{"time":1000,"type":"log","source":"1","event-data":{"logMessage":"again"}},"id":0}'
    # the next step takes the next id
    ctl.send({"command": "time-set", "id": 16, "parameters": {"time": 2000}})
    assert emu.line().startswith(b'{"command":"time-step","id":1002,"parameters":{"time":2000,')
    assert emu2.line().startswith(b'{"command":"time-step","id":1002,')
    emu.send({"reply": "OK", "id": 1002})
    emu2.send({"reply": "OK", "id": 1002})
    assert ctl.line() == b'{"reply":"OK","id":16}'
    for p in (ctl, emu, emu2):
        p.close()


def test_step_message_in_flight_when_the_next_step_begins(server):
    """The node-info array of a time-step message goes out of the connection's own copy of it, not through the output buffer
    (rsim_server.cpp, Connection::stepMark).  An emulator that answers a step from the message's first bytes and reads the
    rest later: the next step begins while most of the first message is unsent -- its rest is set aside before the array
    is touched -- and what is written after a step message (a command reply, a log event) comes after it on the wire."""
    ctl = connect(server)
    emu = Peer.__new__(Peer)
    emu.s = socket.socket()
    emu.s.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 4096)   # (before connect: a small window keeps the server's send short)
    emu.s.settimeout(10)
    emu.s.connect(("127.0.0.1", server))
    emu.s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
    emu.buf = b""
    assert emu.line() + b"\r\n" == GREETING
    n = 60_000
    emu.raw("".join('{"command":"node-config-set","parameters":{"node-id":%d,"position":[%d.5,2.0]}}' % (i + 1, i) for i in range(n)))
    emu.send({"command": "subscribe-event", "id": 1})
    assert emu.line(30.0) == b'{"id":1,"reply":"OK"}'
    objs = ",".join('{"node-id":"%d","rssi":-99.99,"receiving":0,"wireless-channel":26}' % (i + 1) for i in range(n)).encode()
    ctl.send({"command": "time-set", "id": 10, "parameters": {"time": 1000}})
    head = b'{"command":"time-step","id":1001,"parameters":{"time":1000,"node-info":['
    while len(emu.buf) < len(head):
        emu.buf += emu.s.recv(4096)
    assert emu.buf.startswith(head)
    emu.send({"reply": "OK", "id": 1001})                # answered from the first bytes; megabytes of the message are unsent
    assert ctl.line() == b'{"reply":"OK","id":10}'
    emu.send({"command": "log", "parameters": {"node-id": 1, "message": "between"}})   # goes behind the step message
    time.sleep(0.3)                                       # (two sockets: the server is to see the log first)
    ctl.send({"command": "time-set", "id": 11, "parameters": {"time": 2000}})           # ... and so does the next step
    time.sleep(0.2)
    assert emu.line(60.0) == head + objs + b"]}}"
    assert emu.line() == b'{"event":{"time":1000,"type":"log","source":"1","event-data":{"logMessage":"between"}},"id":0}'
    assert emu.line(60.0) == head.replace(b"1001", b"1002").replace(b'"time":1000', b'"time":2000') + objs + b"]}}"
    emu.send({"reply": "OK", "id": 1002})
    assert ctl.line() == b'{"reply":"OK","id":11}'
    ctl.close()
    emu.close()


def test_node_info_objects_that_change_their_length(server):
    """The node-info array of a connection is kept as it was sent last, in pieces of 64 objects, and a changed node's
    fields are rewritten where they lie (rsim_server.cpp, Connection::infoPieces): objects that grow and shrink at the
    beginning, the end and the seams of the pieces, step after step -- the message is what writing every object afresh gives."""
    ctl, emu = connect(server), connect(server)
    n = 150
    chan = {i: 26 for i in range(1, n + 1)}
    state = {i: 0 for i in range(1, n + 1)}
    for i in range(1, n + 1):
        emu.send({"command": "node-config-set", "parameters": {"node-id": i, "position": [float(i), 0.0]}})

    def expect(step_id, time_us):
        objs = ",".join('{"node-id":"%d","rssi":-99.99,"receiving":%d,"wireless-channel":%d}' % (i, state[i], chan[i]) for i in range(1, n + 1))
        return ('{"command":"time-step","id":%d,"parameters":{"time":%d,"node-info":[%s]}}' % (step_id, time_us, objs)).encode()

    def step(k):
        emu.send({"command": "time-get", "id": 500 + k})  # (two sockets: the server has read everything above when it answers this)
        assert emu.line().startswith(('{"id":%d,"reply":"OK"' % (500 + k)).encode())
        ctl.send({"command": "time-set", "id": 10 + k, "parameters": {"time": 1000 * (k + 1)}})
        assert emu.line() == expect(1001 + k, 1000 * (k + 1))
        emu.send({"reply": "OK", "id": 1001 + k})
        assert ctl.line() == ('{"reply":"OK","id":%d}' % (10 + k)).encode()

    step(0)
    rounds = [{1: 5, 64: 123456, 65: 7, 128: 11, 150: 2000000000},          # first / last object, both sides of the seams
              {1: 26, 2: 1, 63: 100, 64: 26, 129: -4, 150: 3},
              {i: (i * 7919) % 100000 for i in range(1, n + 1, 3)}]          # every third object, lengths of one to five digits
    for k, change in enumerate(rounds):
        for i, c in change.items():
            emu.send({"command": "node-config-set", "parameters": {"node-id": i, "wireless-channel": c}})
            chan[i] = c
        if k == 1:                                       # a radio switched off reports state 3: same length, in place
            emu.send({"command": "node-config-set", "parameters": {"node-id": 64, "radio-state": "disabled"}})
            state[64] = 3
        step(k + 1)
    emu.send({"command": "node-config-set", "parameters": {"node-id": n + 1, "position": [0.5, 0.5]}})   # a node joins behind them
    chan[n + 1], state[n + 1] = 26, 0
    n += 1
    step(len(rounds) + 1)
    ctl.close()
    emu.close()


def test_framing(server):
    c = connect(server)
    # brace counting: CR / LF between messages, braces and escaped quotes inside strings, nested objects
    c.raw('\r\n  {"command":"log-not","id":1,"x":"}{\\"}"}\r\n{"command":\r\n"time-get","id":2,"o":{"a":{"b":[1,2]}}}')
    assert c.line() == b'{"id":1,"reply":"error","reply-object":{"class":"command-error","description":"unsupported command: log-not"}}'
    assert c.line() == b'{"id":2,"reply":"OK","reply-object":{"time":0}}'
    # length-prefixed payloads with attributes; size 0 is nothing; a negative size falls back to brace counting
    msg = b'{"command":"time-get","id":3}'
    c.raw(str(len(msg)).encode() + b";type=json\r\n" + msg)
    assert c.line() == b'{"id":3,"reply":"OK","reply-object":{"time":0}}'
    c.raw(b"0\n-1\n" + b'{"command":"time-get","id":4}')
    assert c.line() == b'{"id":4,"reply":"OK","reply-object":{"time":0}}'
    # a message split over several writes
    for part in (b'{"comm', b'and":"time', b'-get","i', b'd":5}'):
        c.raw(part)
        time.sleep(0.05)
    assert c.line() == b'{"id":5,"reply":"OK","reply-object":{"time":0}}'
    # a Latin-1 byte in brace mode comes back as its two UTF-8 bytes; a UTF-8 payload by length stays as it is
    c.raw(b'{"command":"caf\xe9","id":6}')
    assert c.line() == ('{"id":6,"reply":"error","reply-object":{"class":"command-error","description":"unsupported command: café"}}'
                        .encode("utf-8"))
    msg = '{"command":"café","id":7}'.encode("utf-8")
    c.raw(str(len(msg)).encode() + b"\n" + msg)
    assert c.line() == ('{"id":7,"reply":"error","reply-object":{"class":"command-error","description":"unsupported command: café"}}'
                        .encode("utf-8"))
    c.close()


@pytest.mark.parametrize("bad", [
    b'{"command":"time-get" "id":1}',                           # not JSON
    b'{"command":5}',                                           # getString on a number
    b'{"command":"time-get","id":1.5}',                         # getLong on 1.5
    b'{"command":"time-get","id":"7"}',                         # getLong on a string
    b'{"reply":1}',                                             # getString on a number
    b'{"command":"transmit","time":0}',                         # no node-id
    b'{"command":"transmit","node-id":1,"time":"now"}',         # time is not a number
    b'{"command":"log","parameters":{"node-id":5,"message":"x"}}',   # log from an unknown node
    b'{"command":"node-config-set","parameters":[1]}',          # parameters is not an object
    b'{"command":"node-config-set","parameters":{"node-id":1,"position":[1,"a"]}}',
    b'{"command":"configuration-set","parameters":{"propagation-option":"n2n-link","matrix-data":[1]}}',  # no number-of-nodes
    b'abc\n',                                                   # a line that is not a size
    b'99999999999\n',                                           # not an int
    b'30000000\n',                                              # too large a payload
    b'3\n[1]',                                                  # a payload that is not an object
])
def test_reader_errors_close_the_connection(server, bad):
    c = connect(server)
    c.raw(bad)
    assert c.closed()
    # the server itself is fine
    d = connect(server)
    d.send({"command": "time-get", "id": 1})
    assert d.line() == b'{"id":1,"reply":"OK","reply-object":{"time":0}}'
    d.close()


def test_closed_emulator_blocks_the_step(server):
    """the reference never removes a closed connection from the emulator list: the step waits for ever"""
    ctl, emu = connect(server), connect(server)
    emu.send({"command": "node-config-set", "id": 1, "parameters": {"node-id": 1}})
    emu.line()
    emu.close()
    time.sleep(0.2)
    ctl.send({"command": "time-set", "id": 2, "parameters": {"time": 10}})
    ctl.send({"command": "time-get", "id": 3})
    assert ctl.line() == b'{"id":3,"reply":"OK","reply-object":{"time":0}}'
    ctl.close()


# ---------------------------------------------------------------- json.hpp: number text
def java_double(x):
    """Double.toString from CPython's shortest repr (an implementation independent of the product's to_chars)"""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Infinity" if x > 0 else "-Infinity"
    if x == 0:
        return "-0.0" if str(x).startswith("-") else "0.0"
    sign, digits, exp = Decimal(repr(abs(x))).as_tuple()
    digits = "".join(map(str, digits)).lstrip("0")
    exp10 = len(digits) - 1 + exp          # scientific exponent
    digits = digits.rstrip("0") or "0"
    if len(digits) == 1:                   # two digits at least, the pair closest to the exact value
        m, e = ("%.1e" % abs(x)).split("e")
        digits, exp10 = m.replace(".", "").rstrip("0"), int(e)
    neg = "-" if x < 0 else ""
    if 1e-3 <= abs(x) < 1e7:
        if exp10 >= 0:
            ip = digits[:exp10 + 1].ljust(exp10 + 1, "0")
            fp = digits[exp10 + 1:] or "0"
            return neg + ip + "." + fp
        return neg + "0." + "0" * (-exp10 - 1) + digits
    return neg + digits[0] + "." + (digits[1:] or "0") + "E" + str(exp10)


KNOWN = {   # java.lang.Double.toString's documented forms
    1.0: "1.0", -100.0: "-100.0", -99.99: "-99.99", 0.001: "0.001", 1e-4: "1.0E-4", 9999999.0: "9999999.0", 1e7: "1.0E7",
    123456.789: "123456.789", 12345678.9: "1.23456789E7", 1e10: "1.0E10", 4.9e-324: "4.9E-324",
    1.7976931348623157e308: "1.7976931348623157E308", 0.1: "0.1", 1e-5: "1.0E-5", 100.0: "100.0", 2.5e-3: "0.0025",
}


def test_json_number_text_and_parser(tmp_path):
    exe = os.path.join(str(tmp_path), "json_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "json_test.cpp")])
    for x, s in KNOWN.items():
        assert java_double(x) == s
    rng = np.random.default_rng(5)
    xs = list(KNOWN) + [0.0, -0.0, float("nan"), float("inf"), float("-inf")]
    xs += list(rng.uniform(-150, 0, 300)) + list(10.0 ** rng.uniform(-12, 12, 300) * rng.choice([-1, 1], 300))
    xs += [float(v) for v in rng.integers(-10**9, 10**9, 100)] + [float(np.float32(v)) for v in rng.uniform(-100, 100, 100)]
    xs += list(np.frombuffer(rng.bytes(8 * 300), dtype=np.float64))
    xs = [float(x) for x in xs]
    lines = ["d %016x" % struct.unpack("<Q", struct.pack("<d", x))[0] for x in xs]
    cases = [("p", '{"a":[1,2.50,-0,1e5,"x\\n\\u00e9\\"",true,false,null],"b":{}}', '{"a":[1,2.50,-0,1e5,"x\\né\\"",true,false,null],"b":{}}'),
             ("p", ' { "k" : "v" , "k" : 2 } ', '{"k":"v","k":2}'), ("p", '{"a":01}', "error"), ("p", '{"a":1,}', "error"),
             ("p", '{"a":.5}', "error"), ("p", '{"a":"\t"}', "error"), ("p", '[1 2]', "error"), ("p", '{} x', "error"),
             ("l", "12", "12"), ("l", "-9223372036854775808", "-9223372036854775808"), ("l", "9223372036854775808", "error"),
             ("l", "1.0", "error"), ("l", "1e3", "error"), ("l", '"5"', "error")]
    lines += ["%s %s" % (k, a) for k, a, _ in cases]
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60).stdout.split("\n")
    for x, got in zip(xs, out):
        full, wire = got.split(" ")
        want = java_double(x)
        assert full == want, (x, got)
        if want in ("NaN", "Infinity", "-Infinity"):
            assert wire == "refused"
        else:
            assert wire == (want[:-2] if want.endswith(".0") else want)     # cutOffPointZero
    for (k, a, want), got in zip(cases, out[len(xs):]):
        assert got == want, (k, a, got)


def test_emulink_session_shapes(server):
    """The messages the reference's own emulator client sends, member for member (mspsim-emulink EmuLink.java:97-231:
    sendInit, sendPacket, serveForever's time-set, the reply to time-step with the id before the reply member, sendLog),
    and what that client needs back: a greeting it can parse as JSON, time-step with "parameters"."time", the
    controller's reply carrying its id, receive / error messages with a string "node-id"."""
    emu = connect(server)
    emu.raw('{"command":"node-config-set","parameters":{"node-id":1,"position":[10.0,20.0,0.0]}}\r\n')
    emu.raw('{"command":"node-config-set","parameters":{"node-id":2,"position":[15.0,20.0,0.0]}}\r\n')
    emu.raw('{"command":"time-set","parameters":{"time":1000},"id":1}\r\n')          # the emulator is the time controller too
    step = json.loads(emu.line())
    assert step["command"] == "time-step" and step["id"] == 1001 and step["parameters"]["time"] == 1000
    assert [i["node-id"] for i in step["parameters"]["node-info"]] == ["1", "2"]
    emu.raw('{"command":"transmit","node-id":1,"packet-data":"0A0B0C","time":0}\r\n')
    # (this server runs without a medium: the reference answers a transmit it cannot evaluate with an error, id-less here)
    assert emu.line() == b'{"reply":"error","reply-object":{"class":"command-error","description":"no radio medium available"}}'
    emu.raw('{"id":1001,"reply":"OK"}\r\n')
    assert emu.line() == b'{"reply":"OK","id":1}'
    emu.raw('{"command":"log","parameters":{"node-id":1,"message":"booted"}}\r\n')
    emu.raw('{"command":"time-set","parameters":{"time":2000},"id":2}\r\n')
    assert json.loads(emu.line())["id"] == 1002
    emu.raw('{"id":1002,"reply":"OK"}\r\n')
    assert emu.line() == b'{"reply":"OK","id":2}'
    emu.close()
