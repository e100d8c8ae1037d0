"""Packet traces, CPU only: the C++ mirror's PcapListener writes exactly the
bytes the reference's PcapExporter would (derived by hand from util/PcapExporter.java:47-91 -- every
field big-endian, linktype 195), the Python reader understands them, and the compact replay trace
round-trips into per-tick rm_tx_record arrays."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_cpp(rsa, tmp_path):
    lib = os.path.dirname(rsa.library_path())
    exe = os.path.join(str(tmp_path), "host_trace_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_trace_test.cpp"),
                           "-L" + lib, "-lradiomedium_hip", "-Wl,-rpath," + lib])
    pcap, trace = os.path.join(str(tmp_path), "t.pcap"), os.path.join(str(tmp_path), "t.rmt")
    out = subprocess.run([exe, pcap, trace], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
    return pcap, trace


def test_pcap_bytes_are_the_reference_dialect(rsa, tmp_path):
    from radio_sim_amd import trace as T
    pcap, _ = _run_cpp(rsa, tmp_path)
    raw = open(pcap, "rb").read()
    want = bytes.fromhex(
        "a1b2c3d4" "0002" "0004" "00000000" "00000000" "00001000" "000000c3"      # header: snaplen 4096, network 195
        "00000000" "000003e8" "00000005" "00000005" "0102030405"                  # t = 1000 us
        "00000003" "0000007b" "00000003" "00000003" "fffe7f"                      # t = 3 s + 123 us
        "00000000" "00000000" "00000000" "00000000")                              # 4294967296 s -> (int) 0; empty payload
    assert raw == want
    pk = T.read_pcap(pcap)
    assert pk[0] == (1000, bytes([1, 2, 3, 4, 5])) and pk[1] == (3000123, bytes([0xFF, 0xFE, 0x7F])) and pk[2] == (0, b"")
    again = os.path.join(str(tmp_path), "again.pcap")
    T.write_pcap(again, pk)
    assert open(again, "rb").read() == raw


def test_compact_trace_round_trip_and_tick_cut(rsa, O, tmp_path):
    from radio_sim_amd import trace as T
    _, trace = _run_cpp(rsa, tmp_path)
    rows = T.read_trace(trace)
    assert len(rows) == 3
    assert rows["time_us"].tolist() == [1000, 3000123, 4294967296000000]
    assert rows["src"].tolist() == [0, 1, 0] and rows["hex_length"].tolist() == [10, 6, 0]
    assert rows["txpower"].tolist() == [0.0, -3.5, 0.0] and rows["channel"].tolist() == [26, 15, 26]
    # a synthetic trace: cut into ticks, records as RadioPacket(node, time, data) would fill them
    rng = np.random.default_rng(1)
    n = 50
    nd = O.NodeTable(n)
    nd.x, nd.y = rng.uniform(0, 100, n), rng.uniform(0, 100, n)
    nd.txprob[::3] = 0.5
    syn = np.zeros(200, dtype=T.TRACE_DTYPE)
    syn["time_us"] = rng.integers(0, 20000, 200)
    syn["src"], syn["hex_length"] = rng.integers(0, n, 200), 2 * rng.integers(5, 127, 200)
    syn["txpower"], syn["channel"] = rng.uniform(-10, 0, 200), 26
    path = os.path.join(str(tmp_path), "syn.rmt")
    T.write_trace(path, syn)
    back = T.read_trace(path)
    assert np.array_equal(back, syn)
    ticks = T.ticks_of(back, 1000)
    assert sum(len(p) for _, p in ticks) == 200
    for t0, part in ticks:
        assert np.all((part["time_us"] >= t0) & (part["time_us"] < t0 + 1000))
        recs = T.records_of(part, nd)
        assert np.array_equal(recs["air_us"], part["hex_length"] * 32) and np.array_equal(recs["x"], nd.x[part["src"]])
        assert np.array_equal(recs["txprob"], nd.txprob[part["src"]]) and np.array_equal(recs["txpower"], part["txpower"])
    # stable inside a tick: call order survives for equal start times
    same = np.zeros(3, dtype=T.TRACE_DTYPE)
    same["time_us"], same["src"] = 500, [7, 3, 5]
    assert T.ticks_of(same)[0][1]["src"].tolist() == [7, 3, 5]
