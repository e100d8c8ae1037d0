"""Packet traces (host logic, numpy only).

* `read_pcap` / `write_pcap`: the reference's pcap dialect (util/PcapExporter.java:47-91: big-endian
  classic pcap, linktype 195 = IEEE 802.15.4, timestamps in microseconds split at 10^6).
* `read_trace` / `write_trace`: the compact replay trace of host/radiomedium.hpp's TraceListener
  (a pcap file does not name the sending node, so it cannot drive the medium again).
* `ticks_of`: cut a trace into simulated ticks; `records_of`: the ticks' rm_tx_record arrays for
  rm_tick_* / rm_batch_run_device, built from a node table as RadioPacket(node, time, data) would
  (RadioPacket.java:46-52, air time 32 us per hex character, :72).
"""
import struct

import numpy as np

from ._lib import TX_RECORD_DTYPE

PCAP_MAGIC, PCAP_LINKTYPE_802154, PCAP_SNAPLEN = 0xA1B2C3D4, 195, 4096
TRACE_MAGIC = b"RMTRACE1"
TRACE_DTYPE = np.dtype([("time_us", "<i8"), ("src", "<i4"), ("hex_length", "<i4"), ("txpower", "<f8"),
                        ("channel", "<i4"), ("zero", "<i4")])
assert TRACE_DTYPE.itemsize == 32


def write_pcap(path, packets):
    """packets: iterable of (time_us, bytes)."""
    with open(path, "wb") as f:
        f.write(struct.pack(">IHHiIII", PCAP_MAGIC, 2, 4, 0, 0, PCAP_SNAPLEN, PCAP_LINKTYPE_802154))
        for t, data in packets:
            f.write(struct.pack(">iiII", int(t // 1000000), int(t % 1000000), len(data), len(data)))
            f.write(bytes(data))


def read_pcap(path):
    """-> list of (time_us, bytes); rejects anything but the reference's dialect."""
    raw = open(path, "rb").read()
    magic, vmaj, vmin, zone, sigfigs, snaplen, net = struct.unpack_from(">IHHiIII", raw, 0)
    if magic != PCAP_MAGIC or (vmaj, vmin) != (2, 4) or net != PCAP_LINKTYPE_802154:
        raise ValueError("not a big-endian pcap 2.4 file of linktype 195")
    out, off = [], 24
    while off < len(raw):
        sec, usec, incl, orig = struct.unpack_from(">iiII", raw, off)
        off += 16
        if off + incl > len(raw):
            raise ValueError("truncated packet record")
        out.append((sec * 1000000 + usec, raw[off:off + incl]))
        off += incl
    return out


def write_trace(path, rows):
    rows = np.ascontiguousarray(rows, dtype=TRACE_DTYPE)
    with open(path, "wb") as f:
        f.write(TRACE_MAGIC)
        f.write(struct.pack("<Q", len(rows)))
        f.write(rows.tobytes())


def read_trace(path):
    raw = open(path, "rb").read()
    if raw[:8] != TRACE_MAGIC:
        raise ValueError("not a radio-medium trace")
    (count,) = struct.unpack_from("<Q", raw, 8)
    rows = np.frombuffer(raw, dtype=TRACE_DTYPE, offset=16)
    if len(rows) != count:
        raise ValueError("trace says %d records, holds %d" % (count, len(rows)))
    return rows


def ticks_of(rows, tick_us=1000):
    """-> [(t_begin_us, rows of that tick)] in time order; a tick owns the frames that START in it
    (the medium evaluates a frame once, when it is transmitted).  Call order is kept inside a tick."""
    if len(rows) == 0:
        return []
    order = np.argsort(rows["time_us"], kind="stable")
    rows = rows[order]
    tick = rows["time_us"] // tick_us
    cuts = np.flatnonzero(np.diff(tick)) + 1
    return [(int(part["time_us"][0] // tick_us) * tick_us, part) for part in np.split(rows, cuts)]


def records_of(rows, nodes):
    """rm_tx_record array for trace rows; `nodes` has x, y, z, txprob arrays (e.g. workload.NodeTable)."""
    r = np.zeros(len(rows), dtype=TX_RECORD_DTYPE)
    src = rows["src"]
    r["x"], r["y"], r["z"] = nodes.x[src], nodes.y[src], nodes.z[src]
    r["txpower"], r["txprob"] = rows["txpower"], nodes.txprob[src]
    r["start_us"], r["air_us"] = rows["time_us"], rows["hex_length"].astype(np.int64) * 32
    r["src"], r["channel"] = src, rows["channel"]
    return r
